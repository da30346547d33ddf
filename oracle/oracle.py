"""oracle/oracle.py -- TEST INFRASTRUCTURE ONLY: ctypes loader for oracle/libugs_oracle.so.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the
product package never does.  Results come back as numpy int64 arrays shaped like the reference's
tensors (reference src/samplers/ugs_sampler/__init__.pyi:11-56).
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "libugs_oracle.so")

EDGE_MODES = {"local": 0, "flat": 1, "global": 2}
BATCH_MODES = {"sample": 0, "graph": 1, "global": 2}


class _Result(C.Structure):
    _fields_ = [("rows", C.c_int64), ("k", C.c_int64), ("total_edges", C.c_int64), ("num_graphs", C.c_int64),
                ("nodes", C.POINTER(C.c_int64)), ("edge_index", C.POINTER(C.c_int64)),
                ("edge_ptr", C.POINTER(C.c_int64)), ("edge_src", C.POINTER(C.c_int64)),
                ("sample_ptr", C.POINTER(C.c_int64))]


def build(force=False):
    src = os.path.join(HERE, "ugs_oracle.c")
    if force or not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(src):
        subprocess.run(["make", "-C", HERE, "libugs_oracle.so"], check=True, capture_output=True)
    return LIB


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(LIB)
        i64p = C.POINTER(C.c_int64)
        L.ugs_oracle_preproc_create.restype = C.c_void_p
        L.ugs_oracle_preproc_create.argtypes = [i64p, C.c_int64, C.c_int64, C.c_int]
        L.ugs_oracle_preproc_free.argtypes = [C.c_void_p]
        L.ugs_oracle_preproc_info.argtypes = [C.c_void_p, i64p, i64p, C.POINTER(C.c_double), C.POINTER(C.c_int),
                                              C.POINTER(C.c_int)]
        L.ugs_oracle_preproc_dump.argtypes = [C.c_void_p] + [C.c_void_p] * 9
        L.ugs_oracle_sample_range.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_int, C.c_int, C.c_int64, C.c_int,
                                              C.POINTER(_Result)]
        L.ugs_oracle_result_free.argtypes = [C.POINTER(_Result)]
        L.ugs_oracle_cache_create.restype = C.c_void_p
        L.ugs_oracle_cache_create.argtypes = [C.c_int64]
        L.ugs_oracle_cache_free.argtypes = [C.c_void_p]
        L.ugs_oracle_cache_stats.argtypes = [C.c_void_p, i64p, i64p, i64p]
        L.ugs_oracle_sample_batch.argtypes = [C.c_void_p, i64p, C.c_int64, i64p, C.c_int64, C.c_int, C.c_int, C.c_int,
                                              C.c_int, C.POINTER(_Result)]
        L.ugs_oracle_stl_order.restype = C.c_int64
        L.ugs_oracle_stl_order.argtypes = [C.POINTER(C.c_int), C.c_int64, C.POINTER(C.c_int)]
        _lib = L
    return _lib


def _i64(a):
    a = np.ascontiguousarray(np.asarray(a, dtype=np.int64))
    return a, a.ctypes.data_as(C.POINTER(C.c_int64))


def _take(ptr, n, shape=None):
    if n <= 0:
        out = np.zeros(0, dtype=np.int64)
    else:
        out = np.ctypeslib.as_array(ptr, shape=(n,)).copy()
    return out.reshape(shape) if shape is not None else out


def _unpack(res, batch):
    rows, k, es = res.rows, res.k, res.total_edges
    nodes = _take(res.nodes, rows * max(k, 0), (rows, max(k, 0)))
    edge_index = _take(res.edge_index, 2 * es, (2, es))
    edge_ptr = _take(res.edge_ptr, rows + 1)
    edge_src = _take(res.edge_src, es)
    if batch:
        sample_ptr = _take(res.sample_ptr, res.num_graphs + 1)
        return nodes, edge_index, edge_ptr, sample_ptr, edge_src
    return nodes, edge_index, edge_ptr, edge_src


class OracleError(RuntimeError):
    pass


_ERR = {-2: "No viable roots available", -3: "edge_src_idx_local out of range", -4: "bad mode"}


class Preproc:
    """CPU restatement of create_preproc (reference src/preproc.cpp:262-284)."""

    def __init__(self, edge_index, num_nodes, k):
        ei, p = _i64(edge_index)
        assert ei.ndim == 2 and ei.shape[0] == 2
        self.E = ei.shape[1]
        self.n = int(num_nodes)
        self.k = int(k)
        self._h = lib().ugs_oracle_preproc_create(p, self.E, self.n, self.k)

    def close(self):
        if getattr(self, "_h", None) and _lib is not None:
            _lib.ugs_oracle_preproc_free(self._h)
        self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def info(self):
        n, nnz, z, nz, hg = C.c_int64(), C.c_int64(), C.c_double(), C.c_int(), C.c_int()
        lib().ugs_oracle_preproc_info(self._h, C.byref(n), C.byref(nnz), C.byref(z), C.byref(nz), C.byref(hg))
        return {"num_nodes": n.value, "num_edges_stored": nnz.value, "Z": z.value,
                "bucket_count_nonzero": nz.value, "has_graphlets": bool(hg.value)}

    def dump(self):
        inf = self.info()
        n, nnz = inf["num_nodes"], inf["num_edges_stored"]
        d = {"indptr": np.zeros(n + 1, np.int64), "indices": np.zeros(nnz, np.int32),
             "edge_col": np.zeros(nnz, np.int32), "order": np.zeros(n, np.int32), "index_of": np.zeros(n, np.int32),
             "suffix_deg": np.zeros(n, np.int32), "bucket_b": np.zeros(n, np.float64),
             "prob": np.zeros(n, np.float64), "alias": np.zeros(n, np.int32)}
        lib().ugs_oracle_preproc_dump(self._h, *[v.ctypes.data_as(C.c_void_p) for v in d.values()])
        d["Z"] = inf["Z"]
        return d

    def sample(self, m, k, edge_mode="local", base_offset=0, seed=42, i_begin=0, i_end=None):
        """CPU restatement of sample() (reference src/sampler.cpp:91-290); rows [i_begin, i_end) of the m rows."""
        if i_end is None:
            i_end = m
        res = _Result()
        rc = lib().ugs_oracle_sample_range(self._h, int(i_begin), int(i_end), int(k), EDGE_MODES[edge_mode],
                                           int(base_offset), int(seed), C.byref(res))
        if rc != 0:
            raise OracleError(_ERR.get(rc, f"oracle error {rc}"))
        out = _unpack(res, batch=False)
        lib().ugs_oracle_result_free(C.byref(res))
        return out


class Cache:
    """The reference's process-global LRU (include/cache.hpp:15-78); default capacity 1000."""

    def __init__(self, capacity=1000):
        self._h = lib().ugs_oracle_cache_create(int(capacity))

    def close(self):
        if getattr(self, "_h", None) and _lib is not None:
            _lib.ugs_oracle_cache_free(self._h)
        self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def stats(self):
        s, h, m = C.c_int64(), C.c_int64(), C.c_int64()
        lib().ugs_oracle_cache_stats(self._h, C.byref(s), C.byref(h), C.byref(m))
        return {"size": s.value, "hits": h.value, "misses": m.value}


def sample_batch(edge_index, ptr, m_per_graph, k, mode="sample", seed=42, cache=None):
    """CPU restatement of sample_batch (reference src/ugs_sampler_batch_extension.cpp:77-299).
    `cache=None` uses a fresh LRU (a fresh process in the reference)."""
    own = cache is None
    if own:
        cache = Cache()
    ei, pe = _i64(edge_index)
    pt, pp = _i64(ptr)
    res = _Result()
    rc = lib().ugs_oracle_sample_batch(cache._h, pe, ei.shape[1], pp, pt.shape[0] - 1, int(m_per_graph), int(k),
                                       BATCH_MODES[mode], int(seed), C.byref(res))
    if rc != 0:
        raise OracleError(_ERR.get(rc, f"oracle error {rc}"))
    out = _unpack(res, batch=True)
    lib().ugs_oracle_result_free(C.byref(res))
    if own:
        cache.close()
    return out


def stl_order(seq):
    """Iteration order of std::unordered_set<int> after inserting `seq` one by one (restated rule)."""
    a = np.ascontiguousarray(np.asarray(seq, dtype=np.int32))
    out = np.zeros(max(len(a), 1), dtype=np.int32)
    c = lib().ugs_oracle_stl_order(a.ctypes.data_as(C.POINTER(C.c_int)), len(a), out.ctypes.data_as(C.POINTER(C.c_int)))
    return out[:c].copy()
