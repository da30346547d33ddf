"""ugs_sampler -- MI355X-native drop-in for the reference's `ugs_sampler` extension module.

Same import name, function names, parameter names, order and defaults as the reference's pybind module
(AniruddhaMandal/SS-GNN src/samplers/ugs_sampler/src/extension.cpp:4-13, stub __init__.pyi:11-56), so
`self.sampler = ugs_sampler.sample_batch` in gps/experiment.py:163-178 keeps working unchanged.  Calls go
through the C ABI of include/ugs_mi355.h into hand-written HIP kernels; sampling never runs on the CPU.

Additions that the reference does not have (all optional, keyword-only or separate functions):
  * `device=` on sample / sample_batch: return the tensors on that GPU instead of pinned host memory, so the
    trainer's later `batch.to(device)` (gps/experiment.py:523) moves nothing;
  * clear_cache() / cache_stats(): the process-global preprocessing LRU (reference: UGS_CACHE_SIZE, default 1000);
  * Plan: a batch preprocessed once and kept resident in HBM, sampled repeatedly / over row sub-ranges
    (multi-GPU sharding, see ugs_sampler.distributed).
"""
import ctypes as C
import os

import torch

from ._lib import UGS_E_CAPACITY, check, lib, vp

__version__ = (lib.ugs_version() or b"").decode()
__all__ = ["sample", "create_preproc", "destroy_preproc", "has_graphlets", "get_preproc_info", "sample_batch",
           "clear_cache", "cache_stats", "preproc_dump", "Plan", "device_count"]

_EDGE_MODES = {"local": 0, "flat": 1, "global": 2}
_BATCH_MODES = {"sample": 0, "graph": 1, "global": 2}
_FILL_MODES = dict(_BATCH_MODES, batch=3)      # Plan.fill only: row * k + local index (the encoder's batch-level edge index)
_I32_MIN, _I32_MAX = -(2 ** 31), 2 ** 31 - 1


def _as_c_int(x, name):
    """pybind11 rejects Python ints that do not fit the C `int` parameter (TypeError); so do we."""
    if isinstance(x, bool) or not isinstance(x, int):
        try:
            x = x.__index__()
        except Exception:
            raise TypeError(f"{name} must be an integer") from None
    if not (_I32_MIN <= x <= _I32_MAX):
        raise TypeError(f"{name}={x} does not fit a C int")
    return int(x)


def _check_cpu_i64(t, name):
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name} must be a torch.Tensor")
    if t.device.type != "cpu":
        raise RuntimeError(f"{name} must be on CPU")
    if t.dtype != torch.int64:
        raise RuntimeError(f"{name} must be int64")


def _edge_index_view(edge_index):
    """(tensor kept alive, data pointer, row stride in elements, number of columns) of an int64 [2, E] tensor."""
    if edge_index.dim() != 2 or edge_index.size(0) != 2:
        raise RuntimeError("edge_index must have shape [2, E]")
    if edge_index.size(1) > 0 and edge_index.stride(1) != 1:
        edge_index = edge_index.contiguous()
    stride = edge_index.stride(0) if edge_index.size(1) > 0 else 0
    return edge_index, edge_index.data_ptr(), stride, edge_index.size(1)


def _out_opts(device):
    if device is None:
        return dict(dtype=torch.int64, device="cpu", pin_memory=torch.cuda.is_available()), 0
    dev = torch.device(device)
    if dev.type != "cuda":
        raise RuntimeError("device= must be a GPU device (or None for pinned host tensors)")
    return dict(dtype=torch.int64, device=dev), 1


def _carve(opts, shapes):
    """int64 tensors of the given shapes as views of ONE allocation, in order: one allocator call instead of len(shapes), and
    the library moves adjacent outputs with one copy (ugs_host.cpp finish_common).  Each view is contiguous."""
    sizes = [int(torch.Size(s).numel()) for s in shapes]
    buf = torch.empty((sum(sizes),), **opts)
    out, off = [], 0
    for s, n in zip(shapes, sizes):
        out.append(buf[off:off + n].view(s))
        off += n
    return out


def device_count():
    n = C.c_int()
    check(lib.ugs_device_count(C.byref(n)))
    return n.value


def _select_device(device, jobs=False):
    """Device of the calling thread's next library calls.  For a job (`jobs`) that returns DEVICE tensors, the job also runs on
    torch's current stream of that device: the outputs come from torch's stream-ordered allocator, and only stream order keeps
    the job's writes behind kernels that may still read a recycled block."""
    if device is not None:
        idx = torch.device(device).index
        idx = idx if idx is not None else torch.cuda.current_device()
        check(lib.ugs_set_device(idx))
        if jobs:
            check(lib.ugs_set_stream(torch.cuda.current_stream(idx).cuda_stream, 1))
        else:
            check(lib.ugs_set_stream(None, 0))       # only a device job runs on the caller's stream: a stale handle must not outlive it
    else:
        if torch.cuda.is_available():
            check(lib.ugs_set_device(torch.cuda.current_device()))
        check(lib.ugs_set_stream(None, 0))


# ---------------------------------------------------------------------------------------------------------
# handle API
# ---------------------------------------------------------------------------------------------------------
def create_preproc(edge_index, num_nodes, k):
    """Create preprocessing for a graph and return a handle (int)."""
    _check_cpu_i64(edge_index, "edge_index")
    keep, p, stride, e = _edge_index_view(edge_index)
    h = C.c_int64()
    check(lib.ugs_create_preproc(p, stride, e, int(num_nodes), _as_c_int(k, "k"), C.byref(h)))
    return h.value


def destroy_preproc(handle):
    """Destroy a preprocessing handle."""
    check(lib.ugs_destroy_preproc(int(handle)))


def has_graphlets(handle):
    """Return true if preprocessed graph contains k-graphlets."""
    out = C.c_int()
    check(lib.ugs_has_graphlets(int(handle), C.byref(out)))
    return bool(out.value)


def get_preproc_info(handle):
    """Return small metadata dict for debugging."""
    found, n, nnz, z, nz = C.c_int(), C.c_int64(), C.c_int64(), C.c_double(), C.c_int()
    check(lib.ugs_get_preproc_info(int(handle), C.byref(found), C.byref(n), C.byref(nnz), C.byref(z), C.byref(nz)))
    if not found.value:
        return {}
    return {"num_nodes": n.value, "num_edges_stored": nnz.value, "Z": z.value, "bucket_count_nonzero": nz.value}


def preproc_dump(handle):
    """Internals of a handle as numpy arrays (testing aid; not part of the reference surface)."""
    import numpy as np
    info = get_preproc_info(handle)
    if not info:
        raise RuntimeError("Invalid preproc handle")
    n, nnz = info["num_nodes"], info["num_edges_stored"]
    d = {"indptr": np.zeros(n + 1, np.int64), "indices": np.zeros(nnz, np.int32), "edge_col": np.zeros(nnz, np.int32),
         "order": np.zeros(n, np.int32), "index_of": np.zeros(n, np.int32), "suffix_deg": np.zeros(n, np.int32),
         "bucket_b": np.zeros(n, np.float64), "prob": np.zeros(n, np.float64), "alias": np.zeros(n, np.int32)}
    check(lib.ugs_preproc_dump(int(handle), *[v.ctypes.data for v in d.values()]))
    d["Z"] = info["Z"]
    return d


def sample(handle, m_per_graph, k, edge_mode="local", base_offset=0, seed=42, *, device=None):
    """Sample m_per_graph subgraphs of size k from preprocessed graph. edge_mode in {'local', 'flat', 'global'}

    Returns (nodes_t [m,k], edge_index_t [2,Es], edge_ptr_t [m+1], edge_src_t [Es]), all int64."""
    m, k, seed = _as_c_int(m_per_graph, "m_per_graph"), _as_c_int(k, "k"), _as_c_int(seed, "seed")
    if edge_mode not in _EDGE_MODES:
        raise RuntimeError("edge_mode must be one of: 'local', 'flat', 'global'")
    _select_device(device, jobs=True)
    key = ("handle", int(handle), m, k, edge_mode)
    streamed = device is None and m >= _STREAM_MIN_ROWS and not os.environ.get("UGS_NO_STREAMED_CALL")
    if streamed and key in _stream_totals and torch.cuda.is_available():     # large host-visible call of a shape seen before (see sample_batch)
        seen = _stream_totals[key]
        cap = seen + seen // 50 + 4096
        opts, _ = _out_opts(None)
        sizes = [m * k, m + 1, cap, 2 * cap]
        buf = torch.empty((sum(sizes),), **opts)
        o1, o2, o3 = sizes[0], sizes[0] + sizes[1], sizes[0] + sizes[1] + sizes[2]
        base, total = buf.data_ptr(), C.c_int64()
        rc = lib.ugs_sample_stream(int(handle), m, k, _EDGE_MODES[edge_mode], int(base_offset), seed, cap,
                                   base, base + 8 * o3, base + 8 * o1, base + 8 * o2, C.byref(total))
        if rc != UGS_E_CAPACITY:
            check(rc)
            t = total.value
            _stream_totals[key] = max(seen, t)
            return buf[:o1].view(m, k), buf[o3:o3 + 2 * t].view(2, t), buf[o1:o2], buf[o2:o2 + t]
    job, total = vp(), C.c_int64()
    check(lib.ugs_sample_begin(int(handle), m, k, _EDGE_MODES[edge_mode], int(base_offset), seed, C.byref(job), C.byref(total)))
    if streamed:
        if key not in _stream_totals and len(_stream_totals) >= 1024:
            _stream_totals.clear()
        _stream_totals[key] = max(_stream_totals.get(key, 0), total.value)
    try:
        opts, on_dev = _out_opts(device)
        nodes, edge_ptr, edge_index, edge_src = _carve(opts, [(m, k), (m + 1,), (2, total.value), (total.value,)])
    except BaseException:
        lib.ugs_job_cancel(job)
        raise
    check(lib.ugs_sample_finish(job, nodes.data_ptr(), edge_index.data_ptr(), edge_ptr.data_ptr(), edge_src.data_ptr(), on_dev))
    return nodes, edge_index, edge_ptr, edge_src


# ---------------------------------------------------------------------------------------------------------
# batch API
# ---------------------------------------------------------------------------------------------------------
# Large host-visible calls: a call whose shape (columns, graphs, m, k, mode) was seen before hands its pinned output buffers to the
# library up front -- the edge buffers sized by the largest total seen for that shape plus 2 % -- and the library copies finished row
# chunks out while the next ones walk (ugs_sample_batch_stream).  The first call of a shape, and a call that outgrows the estimate,
# take the two-phase path (walk, allocate by the total, fill, copy).  Same tensors either way.
_STREAM_MIN_ROWS = 262144
_stream_totals = {}


def _sample_batch_streamed(p, stride, e, ptr_c, G, m, k, mode, seed):
    key = (e, G, m, k, mode)
    seen = _stream_totals.get(key)
    if seen is None or not torch.cuda.is_available():
        return None
    cap = seen + seen // 50 + 4096
    B = G * m
    opts, _ = _out_opts(None)
    sizes = [B * k, B + 1, G + 1, cap, 2 * cap]
    buf = torch.empty((sum(sizes),), **opts)
    offs = [0]
    for n in sizes:
        offs.append(offs[-1] + n)
    base = buf.data_ptr()
    total = C.c_int64()
    rc = lib.ugs_sample_batch_stream(p, stride, e, ptr_c.data_ptr(), G, m, k, _BATCH_MODES[mode], seed, cap,
                                     base + 8 * offs[0], base + 8 * offs[4], base + 8 * offs[1], base + 8 * offs[2], base + 8 * offs[3],
                                     C.byref(total))
    if rc == UGS_E_CAPACITY:
        return None                      # (the two-phase path repeats the call and raises the estimate)
    check(rc)
    t = total.value
    _stream_totals[key] = max(seen, t)
    return (buf[offs[0]:offs[1]].view(B, k), buf[offs[4]:offs[4] + 2 * t].view(2, t), buf[offs[1]:offs[2]],
            buf[offs[2]:offs[3]], buf[offs[3]:offs[3] + t])


def sample_batch(edge_index, ptr, m_per_graph, k, mode="sample", seed=42, *, device=None):
    """Sample m_per_graph k-subgraphs per graph from a batched PyG edge_index + ptr.

    Returns (nodes_t [B,k], edge_index_t [2,Es], edge_ptr_t [B+1], sample_ptr_t [G+1], edge_src_global_t [Es]),
    B = num_graphs * m_per_graph, all int64 -- pinned host tensors like the reference, or on `device` if given."""
    _check_cpu_i64(edge_index, "edge_index")
    _check_cpu_i64(ptr, "ptr")
    if mode not in _BATCH_MODES:
        raise RuntimeError("mode must be one of: 'sample', 'graph', 'global'")
    m, k, seed = _as_c_int(m_per_graph, "m_per_graph"), _as_c_int(k, "k"), _as_c_int(seed, "seed")
    keep, p, stride, e = _edge_index_view(edge_index)
    ptr_c = ptr.contiguous()
    G = ptr_c.numel() - 1
    _select_device(device, jobs=True)
    if device is None and max(G, 0) * m >= _STREAM_MIN_ROWS and not os.environ.get("UGS_NO_STREAMED_CALL"):
        out = _sample_batch_streamed(p, stride, e, ptr_c, G, m, k, mode, seed)
        if out is not None:
            return out
    job, total = vp(), C.c_int64()
    check(lib.ugs_sample_batch_begin(p, stride, e, ptr_c.data_ptr(), G, m, k, _BATCH_MODES[mode], seed, C.byref(job), C.byref(total)))
    if device is None and max(G, 0) * m >= _STREAM_MIN_ROWS:
        key = (e, G, m, k, mode)
        if key not in _stream_totals and len(_stream_totals) >= 1024:      # (a stream of ever-new shapes: forget, the next calls re-learn)
            _stream_totals.clear()
        _stream_totals[key] = max(_stream_totals.get(key, 0), total.value)
    try:
        opts, on_dev = _out_opts(device)
        B = max(G, 0) * m
        nodes, edge_ptr, edge_index_t, edge_src, sample_ptr = _carve(opts, [(B, k), (B + 1,), (2, total.value), (total.value,), (max(G, 0) + 1,)])
    except BaseException:
        lib.ugs_job_cancel(job)
        raise
    check(lib.ugs_sample_batch_finish(job, nodes.data_ptr(), edge_index_t.data_ptr(), edge_ptr.data_ptr(),
                                      sample_ptr.data_ptr(), edge_src.data_ptr(), on_dev))
    return nodes, edge_index_t, edge_ptr, sample_ptr, edge_src


class GraphStep:
    """Plan.graph_step(): nodes [rows,k], edge_ptr [rows+1], edge_index [2,capacity], edge_src [capacity] live on the device and
    are overwritten by every launch(seed); `result()` trims the edge arrays to the step's total (one host sync)."""

    def __init__(self, plan, m, mode, row_begin, row_count, edge_capacity):
        dev = torch.device("cuda", torch.cuda.current_device())
        k = plan.k
        if edge_capacity is None:
            # every ordered pair of every row, twice: PyG batches store both directions of an edge and the sampler symmetrises
            # every column (reference src/preproc.cpp:32-86), so each pair shows up two times.  Repeated columns can exceed
            # this; the fill kernels never write past the capacity and result() reports the overflow.
            edge_capacity = max(1, 2 * row_count * k * (k - 1))
        self.plan, self.rows, self.capacity = plan, row_count, int(edge_capacity)
        self.nodes = torch.empty((row_count, k), dtype=torch.int64, device=dev)
        self.edge_ptr = torch.empty((row_count + 1,), dtype=torch.int64, device=dev)
        self.edge_index = torch.empty((2, self.capacity), dtype=torch.int64, device=dev)
        self.edge_src = torch.empty((self.capacity,), dtype=torch.int64, device=dev)
        h = vp()
        code = _BATCH_MODES[mode] if mode in _BATCH_MODES else _EDGE_MODES[mode]
        check(lib.ugs_plan_graph_create(plan._h, m, k, code, 0, row_begin, row_count, self.nodes.data_ptr(), self.edge_ptr.data_ptr(),
                                        self.edge_index.data_ptr(), self.capacity, self.edge_src.data_ptr(), C.byref(h)))
        self._h = h

    def launch(self, seed):
        check(lib.ugs_plan_graph_launch(self._h, _as_c_int(seed, "seed"), torch.cuda.current_stream().cuda_stream))
        return self

    def result(self):
        total = int(self.edge_ptr[-1].item())
        if total > self.capacity:
            raise RuntimeError(f"edge capacity {self.capacity} too small for {total} edge entries")
        return self.nodes, self.edge_index[:, :total], self.edge_ptr, self.edge_src[:total]

    def close(self):
        if getattr(self, "_h", None):
            lib.ugs_plan_graph_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def clear_cache():
    """Drop the preprocessing LRU and the cached device plans (what a fresh process has)."""
    check(lib.ugs_cache_clear())


def batch_pass_stats():
    """{"device_plans": batches whose slicing / keys / CSR were built by the device pass, "general_path": batches the host path served}"""
    a, b = C.c_int64(), C.c_int64()
    check(lib.ugs_batch_pass_stats(C.byref(a), C.byref(b)))
    return {"device_plans": a.value, "general_path": b.value}


def cache_stats():
    s, h, m = C.c_int64(), C.c_int64(), C.c_int64()
    check(lib.ugs_cache_stats(C.byref(s), C.byref(h), C.byref(m)))
    return {"size": s.value, "hits": h.value, "misses": m.value}


# ---------------------------------------------------------------------------------------------------------
# device-resident plans
# ---------------------------------------------------------------------------------------------------------
class Plan:
    """A batch (edge_index, ptr) or one preprocessing handle, preprocessed once and resident in HBM.

    sample_rows() produces any contiguous range of the G*m result rows as DEVICE tensors on the caller's current
    stream: row b = g*m + i depends only on (graph g, seed, i), so ranks of a multi-GPU job take disjoint ranges."""

    def __init__(self, _h, num_graphs, k):
        self._h, self.num_graphs, self.k = _h, num_graphs, k

    @classmethod
    def from_batch(cls, edge_index, ptr, k, device=None):
        _check_cpu_i64(edge_index, "edge_index")
        _check_cpu_i64(ptr, "ptr")
        keep, p, stride, e = _edge_index_view(edge_index)
        ptr_c = ptr.contiguous()
        _select_device(device)
        h = vp()
        check(lib.ugs_plan_create_batch(p, stride, e, ptr_c.data_ptr(), ptr_c.numel() - 1, _as_c_int(k, "k"), C.byref(h)))
        return cls(h, ptr_c.numel() - 1, k)

    @classmethod
    def from_handle(cls, handle, k, device=None):
        _select_device(device)
        h = vp()
        check(lib.ugs_plan_create_handle(int(handle), C.byref(h)))
        return cls(h, 1, k)

    def close(self):
        if getattr(self, "_h", None):
            lib.ugs_plan_release(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def twin(self):
        """A second Plan over the same device arrays with scratch of its own.  Calls on one plan are serialised through its
        scratch; sampling consecutive batches through a plan and its twin on two streams keeps two steps in flight, so that the
        next batch's walk fills the tail (and the small kernels) of this one's."""
        h = vp()
        check(lib.ugs_plan_twin(self._h, self.k, C.byref(h)))
        return Plan(h, self.num_graphs, self.k)

    def graph_roots(self, graph, capacity):
        """What the walk kernels read for one graph of the plan, copied back from HBM (testing aid): level, and per order position
        prob / alias / v_self / v_alias (level 0) or the viable list (levels 1, 2)."""
        import numpy as np
        lvl, n, nvia = C.c_int32(), C.c_int32(), C.c_int32()
        d = {"prob": np.zeros(capacity, np.float64), "alias": np.zeros(capacity, np.int32), "v_self": np.zeros(capacity, np.int32),
             "v_alias": np.zeros(capacity, np.int32), "viable_vi": np.zeros(capacity, np.int32), "viable_v": np.zeros(capacity, np.int32)}
        check(lib.ugs_plan_graph_roots(self._h, int(graph), int(capacity), C.byref(lvl), C.byref(n), C.byref(nvia), *[v.ctypes.data for v in d.values()]))
        d.update(level=lvl.value, num_nodes=n.value, num_viable=nvia.value)
        return d

    def info(self):
        g, nv, nnz, nb, tier = C.c_int64(), C.c_int64(), C.c_int64(), C.c_int64(), C.c_int()
        check(lib.ugs_plan_info(self._h, self.k, C.byref(g), C.byref(nv), C.byref(nnz), C.byref(nb), C.byref(tier)))
        return {"num_graphs": g.value, "num_vertices": nv.value, "nnz": nnz.value, "device_bytes": nb.value, "tier": tier.value}

    def last_launch(self):
        name = C.create_string_buffer(128)
        grid, block, lds, ovf = C.c_int(), C.c_int(), C.c_int(), C.c_int64()
        check(lib.ugs_plan_last_launch(self._h, name, 128, C.byref(grid), C.byref(block), C.byref(lds), C.byref(ovf)))
        return {"kernel": name.value.decode(), "grid": grid.value, "block": block.value, "lds_bytes": lds.value, "overflow_rows": ovf.value}

    def set_walk_share(self, percent):
        """Let the walk kernels occupy only `percent` of every CU's resident-block capacity, so that kernels on other streams (the
        collation of the previous batch, RCCL) run beside a walk instead of behind it."""
        check(lib.ugs_plan_set_walk_share(self._h, int(percent)))

    def set_timing(self, on=True):
        check(lib.ugs_plan_set_timing(self._h, 1 if on else 0))

    def get_timing(self):
        """{"walk": (ms, launches), "scan": (...), "fill": (...)} since the last call (HIP events on the launch stream)."""
        ms, n = (C.c_double * 3)(), (C.c_int64 * 3)()
        check(lib.ugs_plan_get_timing(self._h, ms, n))
        return {name: (ms[i], n[i]) for i, name in enumerate(("walk", "scan", "fill"))}

    def walk(self, m_per_graph, mode="sample", seed=42, row_begin=0, row_count=None, extra_node_offset=0, out=None, sync=True):
        """Walk phase: returns (nodes [rows,k], edge_ptr [rows+1], total_edges or None) as device tensors."""
        m, seed = _as_c_int(m_per_graph, "m_per_graph"), _as_c_int(seed, "seed")
        if row_count is None:
            row_count = self.num_graphs * m - row_begin
        dev = torch.device("cuda", torch.cuda.current_device())
        if out is None:
            nodes = torch.empty((row_count, self.k), dtype=torch.int64, device=dev)
            edge_ptr = torch.empty((row_count + 1,), dtype=torch.int64, device=dev)
        else:
            nodes, edge_ptr = out
        total = C.c_int64()
        stream = torch.cuda.current_stream().cuda_stream
        check(lib.ugs_plan_walk(self._h, m, self.k, _BATCH_MODES[mode] if mode in _BATCH_MODES else _EDGE_MODES[mode],
                                int(extra_node_offset), seed, int(row_begin), int(row_count), stream,
                                nodes.data_ptr(), edge_ptr.data_ptr(), C.byref(total) if sync else None))
        return nodes, edge_ptr, (total.value if sync else None)

    def fill(self, m_per_graph, nodes, edge_ptr, total_edges, mode="sample", row_begin=0, extra_node_offset=0, out=None):
        """Fill phase: returns (edge_index [2,total], edge_src [total]) as device tensors."""
        m = _as_c_int(m_per_graph, "m_per_graph")
        row_count = nodes.size(0)
        if out is None:
            edge_index = torch.empty((2, total_edges), dtype=torch.int64, device=nodes.device)
            edge_src = torch.empty((total_edges,), dtype=torch.int64, device=nodes.device)
        else:
            edge_index, edge_src = out
        stream = torch.cuda.current_stream().cuda_stream
        check(lib.ugs_plan_fill(self._h, m, self.k, _FILL_MODES[mode] if mode in _FILL_MODES else _EDGE_MODES[mode],
                                int(extra_node_offset), int(row_begin), int(row_count), stream, nodes.data_ptr(),
                                edge_ptr.data_ptr(), edge_index.data_ptr(), edge_index.stride(0) if edge_index.size(1) else 0,
                                edge_src.data_ptr()))
        return edge_index, edge_src

    def step(self, m_per_graph, mode="sample", seed=42, row_begin=0, row_count=None, extra_node_offset=0, out=None, edge_capacity=None):
        """Walk + fill as one call into edge buffers of a capacity fixed up front (no host read-back in between): returns
        (nodes [rows,k], edge_ptr [rows+1], edge_index [2,capacity], edge_src [capacity]) as device tensors; edge_ptr[-1] is the
        number of valid edge entries.  The hot path's step: for batches of small graphs two launches instead of three."""
        m, seed = _as_c_int(m_per_graph, "m_per_graph"), _as_c_int(seed, "seed")
        if row_count is None:
            row_count = self.num_graphs * m - row_begin
        dev = torch.device("cuda", torch.cuda.current_device())
        if out is None:
            cap = int(edge_capacity if edge_capacity is not None else row_count * self.k * (self.k - 1))
            out = (torch.empty((row_count, self.k), dtype=torch.int64, device=dev), torch.empty((row_count + 1,), dtype=torch.int64, device=dev),
                   torch.empty((2, cap), dtype=torch.int64, device=dev), torch.empty((cap,), dtype=torch.int64, device=dev))
        nodes, edge_ptr, edge_index, edge_src = out
        stream = torch.cuda.current_stream().cuda_stream
        check(lib.ugs_plan_step(self._h, m, self.k, _BATCH_MODES[mode] if mode in _BATCH_MODES else _EDGE_MODES[mode], int(extra_node_offset),
                                seed, int(row_begin), int(row_count), stream, nodes.data_ptr(), edge_ptr.data_ptr(), edge_index.data_ptr(),
                                edge_index.stride(0) if edge_index.size(1) else 0, edge_src.data_ptr()))
        return nodes, edge_ptr, edge_index, edge_src

    def graph_step(self, m_per_graph, mode="sample", row_begin=0, row_count=None, edge_capacity=None):
        """The whole step (walk tiers, scan, fill) for a fixed row range captured once as a HIP graph; `launch(seed)` replays it
        on the current stream into the step's own device tensors.  For launch-bound batches of small graphs."""
        m = _as_c_int(m_per_graph, "m_per_graph")
        if row_count is None:
            row_count = self.num_graphs * m - row_begin
        return GraphStep(self, m, mode, int(row_begin), int(row_count), edge_capacity)

    def encoder_inputs(self, m_per_graph, seed=42, row_begin=0, row_count=None):
        """What the reference's encoder derives from the sampler's output (src/gps/gps/models/ss_gnn.py:441-468), straight from
        the kernels: (nodes clamped at 0 [rows*k], validity mask [rows*k], edge_index numbered row*k + local index [2,Es],
        edge_src [Es], batch vector [rows*k]) as device tensors -- no repeat_interleave (and no host synchronisation for it)
        on the consumer's side."""
        nodes, edge_ptr, total = self.walk(m_per_graph, "sample", seed, row_begin, row_count)
        edge_index, edge_src = self.fill(m_per_graph, nodes, edge_ptr, total, "batch", row_begin)
        flat = nodes.flatten()
        batch = torch.arange(nodes.size(0), device=nodes.device).repeat_interleave(self.k)
        return flat.clamp(min=0), flat >= 0, edge_index, edge_src, batch

    def sample_rows(self, m_per_graph, mode="sample", seed=42, row_begin=0, row_count=None):
        """(nodes, edge_index, edge_ptr, edge_src) for rows [row_begin, row_begin+row_count) as device tensors."""
        nodes, edge_ptr, total = self.walk(m_per_graph, mode, seed, row_begin, row_count)
        edge_index, edge_src = self.fill(m_per_graph, nodes, edge_ptr, total, mode, row_begin)
        return nodes, edge_index, edge_ptr, edge_src
