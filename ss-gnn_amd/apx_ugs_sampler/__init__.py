"""apx_ugs_sampler -- drop-in for the reference's `apx_ugs_sampler` extension module
(AniruddhaMandal/SS-GNN src/samplers/apx_ugs_sampler/src/apx_ugs_sampler.cpp; pybind signature :528-537):
sample_batch(edge_index, ptr, m_per_graph, k, mode="sample", seed=42, epsilon=0.1) -> (samples int64 [k, S], arange(S+1)).

Reference semantics kept: only the FIRST graph is sampled, ptr[0]:ptr[1] is a range of edge columns, failed samples are
dropped (S <= m_per_graph), `mode` is accepted and ignored.  The reference draws everything from one sequential
std::mt19937_64 stream, so the entry point is a host computation (C ABI ugs_apx_sample_batch) that is bit-exact with it on
the same toolchain; it is not part of the GPU hot path.
"""
import ctypes as C

import torch

from ugs_sampler._lib import check, lib

__all__ = ["sample_batch"]


def sample_batch(edge_index, ptr, m_per_graph, k, mode="sample", seed=42, epsilon=0.1, *, backend="host", return_order=False):
    """APX-UGS epsilon-uniform graphlet sampling

    backend="host" (default): the reference's sequential generator, bit-exact with the reference.
    backend="gpu": the same algorithm on the GPU, one generator per (sample, trial), all samples and trials side by side; same
    output law (statistical parity), deterministic in (graph, seed); 2 <= k <= 8.  return_order=True (gpu only) appends the
    APX-DD order positions and bucket estimates it used (testing aid)."""
    if backend not in ("host", "gpu"):
        raise RuntimeError("backend must be 'host' or 'gpu'")
    ei = edge_index.cpu().to(torch.int64)
    if ei.size(1) > 0 and ei.stride(1) != 1:
        ei = ei.contiguous()
    pt = ptr.cpu().to(torch.int64).contiguous()
    m, k = int(m_per_graph), int(k)
    out = torch.zeros((max(m, 0), k), dtype=torch.int64)
    n = C.c_int64()
    if backend == "gpu":
        nv = int(ei[:, max(int(pt[0]), 0):int(pt[1])].max().item()) + 1 if ei.size(1) and int(pt[1]) > int(pt[0]) else 0
        pos = torch.full((max(nv, 1),), -1, dtype=torch.int32)
        est = torch.zeros((max(nv, 1),), dtype=torch.float64)
        check(lib.ugs_apx_gpu_sample_batch(ei.data_ptr(), ei.stride(0) if ei.size(1) else 0, ei.size(1), pt.data_ptr(), pt.numel(), m, k,
                                           C.c_uint64(int(seed) & 0xFFFFFFFFFFFFFFFF), C.c_double(float(epsilon)), out.data_ptr(), C.byref(n),
                                           pos.data_ptr(), est.data_ptr(), pos.numel()))
        s = n.value
        res = (out[:s].t().contiguous(), torch.arange(0, s + 1, dtype=torch.int64)) if s else \
              (torch.zeros((k, 0), dtype=torch.int64), torch.zeros((1,), dtype=torch.int64))
        return res + (pos[:nv], est[:nv]) if return_order else res
    check(lib.ugs_apx_sample_batch(ei.data_ptr(), ei.stride(0) if ei.size(1) else 0, ei.size(1), pt.data_ptr(), pt.numel(), m, k,
                                   C.c_uint64(int(seed) & 0xFFFFFFFFFFFFFFFF), C.c_double(float(epsilon)), out.data_ptr(), C.byref(n)))
    s = n.value
    if s == 0:
        return torch.zeros((k, 0), dtype=torch.int64), torch.zeros((1,), dtype=torch.int64)
    return out[:s].t().contiguous(), torch.arange(0, s + 1, dtype=torch.int64)
