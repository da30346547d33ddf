"""epsilon_uniform_sampler -- MI355X-native drop-in for the reference's `epsilon_uniform_sampler` extension module
(AniruddhaMandal/SS-GNN src/samplers/epsilon_uniform_sampler/src/epsilon_uniform_sampler.cpp:122-377; pybind signature
:366-377): sample_batch(edge_index, ptr, m_per_graph, k, mode="sample", seed=42, epsilon=0.1) -> the 5-tuple
(nodes_t, edge_index_t, edge_ptr_t, sample_ptr_t, edge_src_t), int64, returned on the device of `edge_index`.

Sampling runs in HIP kernels (one lane per sample, counter-based generator per (row, attempt)).  The reference is not
deterministic for this sampler (per-thread generators seeded with the OpenMP thread id, dynamic schedule, rows written in
completion order); this implementation is deterministic in (seed, row) and its parity with the reference is statistical:
same growth law, same acceptance law min(1, eps/(w+eps)), same attempt budget max(10, 10/eps), same output format.
"""
import ctypes as C

import torch

from ugs_sampler._lib import check, lib, vp

__all__ = ["sample_batch"]


def sample_batch(edge_index, ptr, m_per_graph, k, mode="sample", seed=42, epsilon=0.1):
    """Epsilon-uniform connected subgraph sampling via random walk with rejection sampling"""
    if edge_index.dtype != torch.int64:
        raise RuntimeError("edge_index must be int64")
    if ptr.dtype != torch.int64:
        raise RuntimeError("ptr must be int64")
    if not (epsilon > 0.0 and epsilon <= 1.0):
        raise RuntimeError("epsilon must be in (0, 1]")
    in_dev = edge_index.device
    ei = edge_index.cpu()
    if ei.dim() != 2 or ei.size(0) != 2:
        raise RuntimeError("edge_index must have shape [2, E]")
    if ei.size(1) > 0 and ei.stride(1) != 1:
        ei = ei.contiguous()
    pt = ptr.cpu().contiguous()
    G = pt.numel() - 1
    m, k = int(m_per_graph), int(k)
    seed = int(seed) & 0xFFFFFFFFFFFFFFFF
    if in_dev.type == "cuda":     # device in, device out: the job runs on torch's current stream of that device (see ugs_set_stream)
        idx = in_dev.index if in_dev.index is not None else torch.cuda.current_device()
        check(lib.ugs_set_device(idx))
        check(lib.ugs_set_stream(torch.cuda.current_stream(idx).cuda_stream, 1))
    else:
        if torch.cuda.is_available():
            check(lib.ugs_set_device(torch.cuda.current_device()))
        check(lib.ugs_set_stream(None, 0))
    job, total = vp(), C.c_int64()
    check(lib.ugs_eps_sample_batch_begin(ei.data_ptr(), ei.stride(0) if ei.size(1) else 0, ei.size(1), pt.data_ptr(), G, m, k,
                                         0 if mode == "sample" else 1, C.c_uint64(seed), C.c_double(float(epsilon)),
                                         C.byref(job), C.byref(total)))
    on_dev = in_dev.type == "cuda"
    try:
        opts = dict(dtype=torch.int64, device=in_dev) if on_dev else dict(dtype=torch.int64, device="cpu", pin_memory=torch.cuda.is_available())
        B = max(G, 0) * m
        nodes = torch.empty((B, k), **opts)
        eidx = torch.empty((2, total.value), **opts)
        eptr = torch.empty((B + 1,), **opts)
        sptr = torch.empty((max(G, 0) + 1,), **opts)
        esrc = torch.empty((total.value,), **opts)
    except BaseException:
        lib.ugs_job_cancel(job)
        raise
    check(lib.ugs_eps_sample_batch_finish(job, nodes.data_ptr(), eidx.data_ptr(), eptr.data_ptr(), sptr.data_ptr(), esrc.data_ptr(), 1 if on_dev else 0))
    return nodes, eidx, eptr, sptr, esrc
