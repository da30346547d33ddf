"""Backend adapters for tests/scenarios.py: the reference (only in the build container), the CPU oracle,
and the HIP product.  All return numpy int64 arrays."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from scenarios import Backend  # noqa: E402


class OracleBackend(Backend):
    """oracle/ugs_oracle.c through oracle/oracle.py (the checker)."""

    def __init__(self, cache_capacity=1000):
        import oracle
        self.o = oracle
        self.cap = cache_capacity
        self.cache = oracle.Cache(cache_capacity)

    def fresh(self):
        self.cache.close()
        self.cache = self.o.Cache(self.cap)

    def create_preproc(self, edge_index, num_nodes, k):
        return self.o.Preproc(edge_index, num_nodes, k)

    def info(self, h):
        return h.info()

    def sample(self, h, m, k, edge_mode, base_offset, seed):
        if h._h is None:
            raise RuntimeError("Invalid preproc handle")
        return h.sample(m, k, edge_mode, base_offset, seed)

    def destroy_preproc(self, h):
        h.close()

    def sample_batch(self, edge_index, ptr, m, k, mode, seed):
        return self.o.sample_batch(edge_index, ptr, m, k, mode, seed, cache=self.cache)


class RefBackend(Backend):
    """The reference pybind module built by oracle/build_ref.py.  Its LRU is process-global and cannot be
    reset: fresh() is a no-op, so drive one scenario per process (oracle/make_golden.py does)."""

    def __init__(self):
        import torch
        import build_ref
        self.torch = torch
        self.m = build_ref.load()

    def _t(self, a):
        return self.torch.from_numpy(np.ascontiguousarray(np.asarray(a, dtype=np.int64)))

    def create_preproc(self, edge_index, num_nodes, k):
        return self.m.create_preproc(self._t(edge_index), int(num_nodes), int(k))

    def info(self, h):
        d = dict(self.m.get_preproc_info(h))
        d["has_graphlets"] = bool(self.m.has_graphlets(h))
        return d

    def sample(self, h, m, k, edge_mode, base_offset, seed):
        return tuple(x.numpy() for x in self.m.sample(h, m, k, edge_mode, base_offset, seed))

    def destroy_preproc(self, h):
        self.m.destroy_preproc(h)

    def sample_batch(self, edge_index, ptr, m, k, mode, seed):
        return tuple(x.numpy() for x in self.m.sample_batch(self._t(edge_index), self._t(ptr), m, k, mode, seed))


class ProductBackend(Backend):
    """The MI355X product: the `ugs_sampler` drop-in package (ctypes -> libugs_mi355.so -> HIP kernels)."""

    def __init__(self):
        import torch
        sys.path.insert(0, os.path.join(ROOT, "ss-gnn_amd"))
        import ugs_sampler
        self.torch = torch
        self.m = ugs_sampler

    def fresh(self):
        self.m.clear_cache()

    def _t(self, a):
        return self.torch.from_numpy(np.ascontiguousarray(np.asarray(a, dtype=np.int64)))

    def create_preproc(self, edge_index, num_nodes, k):
        return self.m.create_preproc(self._t(edge_index), int(num_nodes), int(k))

    def info(self, h):
        d = dict(self.m.get_preproc_info(h))
        d["has_graphlets"] = bool(self.m.has_graphlets(h))
        return d

    def sample(self, h, m, k, edge_mode, base_offset, seed):
        return tuple(x.cpu().numpy() for x in self.m.sample(h, m, k, edge_mode, base_offset, seed))

    def destroy_preproc(self, h):
        self.m.destroy_preproc(h)

    def sample_batch(self, edge_index, ptr, m, k, mode, seed):
        return tuple(x.cpu().numpy() for x in self.m.sample_batch(self._t(edge_index), self._t(ptr), m, k, mode=mode, seed=seed))
