import os, sys, time
sys.path.insert(0, "/root/repo/ss-gnn_amd") if os.path.isdir("/root/repo/ss-gnn_amd") else None
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, os.path.join(ROOT, "ss-gnn_amd"))
import numpy as np, torch, ugs_sampler, ugs_workloads as wl
name = "c3_proteins_b8192"
ei, ptr, m, k = wl.workload(name)
G = len(ptr) - 1; n_per = int(ptr[1] - ptr[0]); cols_per = ei.shape[1] // G
ptr_t = torch.from_numpy(ptr); rng = np.random.default_rng(1)
def mk():
    perm = rng.permutation(G)
    return torch.from_numpy(np.ascontiguousarray(np.concatenate([ei[:, g * cols_per:(g + 1) * cols_per] - g * n_per + i * n_per for i, g in enumerate(perm)], axis=1)))
ugs_sampler.sample_batch(torch.from_numpy(ei), ptr_t, m, k, mode="sample", seed=42)
for _ in range(10): ugs_sampler.sample_batch(mk(), ptr_t, m, k, mode="sample", seed=42)
os.environ["UGS_BP_TRACE"] = "1"
for _ in range(8):
    e = mk(); t = time.perf_counter(); ugs_sampler.sample_batch(e, ptr_t, m, k, mode="sample", seed=42); print("call %.1f us" % ((time.perf_counter() - t) * 1e6), flush=True)
