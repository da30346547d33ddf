"""Stages of the device batch pass when every graph of a call is new to the LRU (UGS_BP_TRACE), PROTEINS-shaped batches."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ss-gnn_amd"))
import numpy as np, torch, ugs_sampler, ugs_workloads as wl
m, k = 256, 6
def fresh(i):
    ei, ptr = wl.tu_batch(39, 73, 32, dataset_seed=7, first_graph=1000 + 32 * i)
    return torch.from_numpy(ei), torch.from_numpy(ptr)
for i in range(45): ugs_sampler.sample_batch(*fresh(i), m, k, mode="sample", seed=42)
os.environ["UGS_BP_TRACE"] = "1"
for i in range(45, 53):
    e, p = fresh(i); t = time.perf_counter(); ugs_sampler.sample_batch(e, p, m, k, mode="sample", seed=42); print("cold call %.1f us" % ((time.perf_counter() - t) * 1e6), flush=True)
