"""Where a drop-in call on a NEW combination of known graphs spends its time (C3 / C4 shapes): device batch pass against the
general host path, host-visible and device outputs; UGS_BP_TRACE=1 prints the stages of the pass."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ss-gnn_amd"))
import numpy as np, torch, ugs_sampler, ugs_workloads as wl
dev = torch.device("cuda:0")
for name in ("c3_proteins_b8192", "c4_qm9_b65536", "c2_mutag_b1024"):
    ei, ptr, m, k = wl.workload(name)
    G = len(ptr) - 1; n_per = int(ptr[1] - ptr[0]); cols_per = ei.shape[1] // G
    ptr_t = torch.from_numpy(ptr); rng = np.random.default_rng(1)
    def mk(count):
        out = []
        for _ in range(count):
            perm = rng.permutation(G)
            out.append(torch.from_numpy(np.ascontiguousarray(np.concatenate([ei[:, g * cols_per:(g + 1) * cols_per] - g * n_per + i * n_per for i, g in enumerate(perm)], axis=1))))
        return out
    ugs_sampler.sample_batch(torch.from_numpy(ei), ptr_t, m, k, mode="sample", seed=42)
    def run(batches, **kw):
        t = time.perf_counter()
        for e in batches: o = ugs_sampler.sample_batch(e, ptr_t, m, k, mode="sample", seed=42, **kw)
        torch.cuda.synchronize()
        return (time.perf_counter() - t) / len(batches) * 1e3
    run(mk(5)); run(mk(5), device=dev)
    # interleaved: the two paths take turns call by call (each call a batch of its own, timed on its own), medians -- a drift of the
    # box or of the allocator cannot favour one of them
    res = {}
    for kw_tag, kw in (("host", {}), ("dev", {"device": dev})):
        ts = {"pass": [], "general": []}
        for e in mk(120):
            tag = "pass" if len(ts["pass"]) <= len(ts["general"]) else "general"
            os.environ["UGS_DEVICE_BATCH"] = "1" if tag == "pass" else "0"
            t = time.perf_counter()
            o = ugs_sampler.sample_batch(e, ptr_t, m, k, mode="sample", seed=42, **kw)
            if kw: torch.cuda.synchronize()
            ts[tag].append(time.perf_counter() - t)
            del o
        for tag in ts:
            v = sorted(ts[tag][5:])
            res.setdefault(tag, []).append(v[len(v) // 2] * 1e3)
    os.environ.pop("UGS_DEVICE_BATCH", None)
    rep = torch.from_numpy(ei)
    t = time.perf_counter()
    for _ in range(20): ugs_sampler.sample_batch(rep, ptr_t, m, k, mode="sample", seed=42)
    rep_ms = (time.perf_counter() - t) / 20 * 1e3
    t = time.perf_counter()
    for _ in range(20): ugs_sampler.sample_batch(rep, ptr_t, m, k, mode="sample", seed=42, device=dev)
    torch.cuda.synchronize(); rep_dev = (time.perf_counter() - t) / 20 * 1e3
    print(f"{name}: new combination host-visible / device-out ms: pass {res['pass'][0]:.3f} / {res['pass'][1]:.3f}   general {res['general'][0]:.3f} / {res['general'][1]:.3f}   repeated batch {rep_ms:.3f} / {rep_dev:.3f}", flush=True)
    if os.environ.get("UGS_BP_TRACE"):
        pass
