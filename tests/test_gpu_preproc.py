"""GPU: device-side preprocessing of large graphs (csrc/ugs_preproc.hip) against the host path and
the CPU oracle.  Everything compared is integer or an exactly reproduced double (weights, Z, alias table): bit-exact."""
import os
import random

import numpy as np
import pytest
import torch

import oracle

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(300)]


@pytest.fixture()
def forced_device():
    old = os.environ.get("UGS_DEVICE_PREPROC")
    os.environ["UGS_DEVICE_PREPROC"] = "1"
    yield
    if old is None:
        os.environ.pop("UGS_DEVICE_PREPROC", None)
    else:
        os.environ["UGS_DEVICE_PREPROC"] = old


def _dump(ei, n, k, mode):
    import ugs_sampler
    os.environ["UGS_DEVICE_PREPROC"] = mode
    h = ugs_sampler.create_preproc(torch.from_numpy(ei), n, k)
    d = ugs_sampler.preproc_dump(h)
    info = ugs_sampler.get_preproc_info(h)
    ugs_sampler.destroy_preproc(h)
    return d, info


def _cases():
    rng = random.Random(11)
    out = []
    for _ in range(40):
        n = rng.choice([1, 2, 3, 7, 20, 64, 200, 500])
        p = rng.choice([0.02, 0.1, 0.4, 0.9])
        k = rng.randint(1, 9)
        e = [(u, v) for u in range(n) for v in range(u + 1, n) if rng.random() < p]
        if rng.random() < 0.5:
            e = e + [(v, u) for u, v in e]
        if rng.random() < 0.5:
            e += [(rng.randrange(n),) * 2 for _ in range(3)]                       # self loops
            e += [(n + 3, 0), (-1, 0), (0, n), (2 ** 40, 0)]                         # out-of-range columns: skipped
            e += e[: len(e) // 3]                                                   # repeated columns (multigraph)
        rng.shuffle(e)
        if not e:
            e = [(0, 0)]
        out.append((np.array(e, dtype=np.int64).T.reshape(2, -1).copy(), n, k))
    # a hub (one long row), a path (reachability needs the full breadth-first search), k = 32
    hub = np.array([[0] * 3000, list(range(1, 3001))], dtype=np.int64)
    out.append((hub, 3001, 5))
    path = np.array([list(range(0, 999)), list(range(1, 1000))], dtype=np.int64)
    out.append((path, 1000, 32))
    out.append((path[:, ::-1].copy(), 1000, 7))
    return out


def test_device_preprocessing_equals_host_and_oracle(forced_device):
    for ei, n, k in _cases():
        dev, idev = _dump(ei, n, k, "1")
        host, ihost = _dump(ei, n, k, "0")
        P = oracle.Preproc(ei, n, k)
        want = P.dump()
        for key in want:
            assert np.array_equal(dev[key], want[key]), (key, n, k, "device vs oracle")
            assert np.array_equal(host[key], want[key]), (key, n, k, "host vs oracle")
        assert idev == ihost == {x: P.info()[x] for x in idev}
        P.close()


def test_sampling_through_a_device_preprocessed_graph_is_bit_exact(forced_device):
    import ugs_sampler
    from ugs_workloads import er_graph
    ei, _ = er_graph(20000, 200000, seed=5)
    n, k, m = 20000, 6, 4000
    os.environ["UGS_DEVICE_PREPROC"] = "1"
    ugs_sampler.clear_cache()
    ptr = torch.tensor([0, n], dtype=torch.long)
    got = ugs_sampler.sample_batch(torch.from_numpy(ei), ptr, m, k, "sample", 77)
    cache = oracle.Cache()
    want = oracle.sample_batch(ei, np.array([0, n], dtype=np.int64), m, k, "sample", 77, cache)
    for g, w in zip(got, want):
        assert np.array_equal(g.numpy(), np.asarray(w))
    cache.close()
    ugs_sampler.clear_cache()


def test_default_threshold_large_graph_equals_host_path():
    """>= 2^21 columns take the device path by default; the result equals the host path's."""
    import ugs_sampler
    from ugs_workloads import er_graph
    os.environ.pop("UGS_DEVICE_PREPROC", None)
    n = 200000
    ei, _ = er_graph(n, 2_200_000, seed=9)
    assert ei.shape[1] >= 1 << 21
    h = ugs_sampler.create_preproc(torch.from_numpy(ei), n, 8)
    d = ugs_sampler.preproc_dump(h)
    ugs_sampler.destroy_preproc(h)
    os.environ["UGS_DEVICE_PREPROC"] = "0"
    h = ugs_sampler.create_preproc(torch.from_numpy(ei), n, 8)
    d0 = ugs_sampler.preproc_dump(h)
    ugs_sampler.destroy_preproc(h)
    os.environ.pop("UGS_DEVICE_PREPROC", None)
    for key in d0:
        assert np.array_equal(d[key], d0[key]), key


def test_plan_assembled_from_device_and_host_pieces(forced_device):
    """One batch whose graphs are partly LRU hits preprocessed on the host and partly new graphs whose CSR is still in HBM:
    the plan's adjacency is written by a kernel for the latter, copied for the former; columns shuffled (non-identity map)."""
    import ugs_sampler
    rng = random.Random(4)
    sizes = [30, 45, 60, 25]
    graphs = []
    for n in sizes:
        e = [(u, v) for u in range(n) for v in range(u + 1, n) if rng.random() < 0.2]
        graphs.append((n, e + [(v, u) for u, v in e]))

    def batch(idx, shuffle):
        cols, ptr = [], [0]
        for i in idx:
            n, e = graphs[i]
            cols += [(u + ptr[-1], v + ptr[-1]) for u, v in e]
            ptr.append(ptr[-1] + n)
        if shuffle:
            rng.shuffle(cols)
        return np.array(cols, dtype=np.int64).T.reshape(2, -1).copy(), np.array(ptr, dtype=np.int64)

    ugs_sampler.clear_cache()
    os.environ["UGS_DEVICE_PREPROC"] = "0"
    ei, ptr = batch([0, 1], False)
    ugs_sampler.sample_batch(torch.from_numpy(ei), torch.from_numpy(ptr), 8, 4, "sample", 1)        # graphs 0, 1 -> LRU (host)
    os.environ["UGS_DEVICE_PREPROC"] = "1"
    for idx, shuffle, mode in (([2, 0, 3, 1], True, "sample"), ([1, 2, 3], False, "global"), ([3, 3, 0], True, "graph")):
        ei, ptr = batch(idx, shuffle)
        got = ugs_sampler.sample_batch(torch.from_numpy(ei), torch.from_numpy(ptr), 50, 4, mode, 9)
        want = oracle.sample_batch(ei, ptr, 50, 4, mode, 9)
        for g, w in zip(got, want):
            assert np.array_equal(g.numpy(), np.asarray(w)), (idx, mode)
    ugs_sampler.clear_cache()
