"""Where a repeated drop-in call spends its time on the host: Python shim against the two C-ABI calls (C3 / C2 shapes)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ss-gnn_amd"))
import numpy as np, torch, ugs_sampler, ugs_workloads as wl
import ugs_sampler as us
lib = us.lib if hasattr(us, "lib") else None
import ctypes as C
for name in ("c3_proteins_b8192", "c2_mutag_b1024"):
    ei, ptr, m, k = wl.workload(name)
    e_t, p_t = torch.from_numpy(ei), torch.from_numpy(ptr)
    for _ in range(20): ugs_sampler.sample_batch(e_t, p_t, m, k, mode="sample", seed=42)
    ts = []
    for _ in range(200):
        t = time.perf_counter(); ugs_sampler.sample_batch(e_t, p_t, m, k, mode="sample", seed=42); ts.append(time.perf_counter() - t)
    ts.sort(); full = ts[len(ts) // 2] * 1e6
    # the two C calls alone, outputs preallocated once
    G = len(ptr) - 1; B = G * m
    job, total = C.c_void_p(), C.c_int64()
    from ugs_sampler import _lib as L
    lb = L.lib
    tb, tf, ta = [], [], []
    for _ in range(200):
        t0 = time.perf_counter()
        rc = lb.ugs_sample_batch_begin(e_t.data_ptr(), e_t.stride(0), e_t.shape[1], p_t.data_ptr(), G, m, k, 0, 42, C.byref(job), C.byref(total))
        t1 = time.perf_counter()
        assert rc == 0
        opts, on_dev = us._out_opts(None)
        nodes, edge_ptr, edge_index_t, edge_src, sample_ptr = us._carve(opts, [(B, k), (B + 1,), (2, total.value), (total.value,), (G + 1,)])
        t2 = time.perf_counter()
        rc = lb.ugs_sample_batch_finish(job, nodes.data_ptr(), edge_index_t.data_ptr(), edge_ptr.data_ptr(), sample_ptr.data_ptr(), edge_src.data_ptr(), on_dev)
        t3 = time.perf_counter()
        assert rc == 0
        tb.append(t1 - t0); ta.append(t2 - t1); tf.append(t3 - t2)
    med = lambda v: sorted(v)[len(v) // 2] * 1e6
    print(f"{name}: sample_batch {full:.1f} us = begin {med(tb):.1f} + allocate {med(ta):.1f} + finish {med(tf):.1f} + shim {full - med(tb) - med(ta) - med(tf):.1f}", flush=True)
