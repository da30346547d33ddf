"""CPU, build container only: the oracle against the reference ITSELF (oracle/_ref, built from
/root/reference by oracle/build_ref.py) on seeded random inputs.  Skipped where the reference is absent
(the GPU box): there the committed fixtures of tests/golden/ carry the same evidence."""
import os
import random

import numpy as np
import pytest

import oracle

REF_SRC = "/root/reference/src/samplers/ugs_sampler"
pytestmark = pytest.mark.skipif(not os.path.isdir(REF_SRC), reason="reference sources not present on this machine")


@pytest.fixture(scope="module")
def ref():
    import torch  # noqa: F401
    import build_ref
    build_ref.build()
    return build_ref.load()


def _rand_graph(rng, n, p, both):
    e = [(u, v) for u in range(n) for v in range(u + 1, n) if rng.random() < p]
    if both:
        e = e + [(v, u) for u, v in e]
    return np.array(e, dtype=np.int64).T.reshape(2, -1)


def test_handle_api_random(ref):
    import torch
    rng = random.Random(2024)
    for _ in range(120):
        n = rng.choice([3, 5, 8, 12, 20, 40, 80, 150])
        ei = _rand_graph(rng, n, rng.choice([0.05, 0.1, 0.2, 0.4, 0.6]), rng.random() < 0.5)
        k, m = rng.randint(1, 8), rng.choice([1, 5, 50])
        seed = rng.choice([42, 0, 7, -5, 123456789])
        mode, bo = rng.choice(["local", "flat", "global"]), rng.choice([0, 10])
        h = ref.create_preproc(torch.from_numpy(ei), n, k)
        P = oracle.Preproc(ei, n, k)
        info, oi = ref.get_preproc_info(h), P.info()
        assert (info["num_nodes"], info["num_edges_stored"], info["Z"], info["bucket_count_nonzero"]) == \
               (oi["num_nodes"], oi["num_edges_stored"], oi["Z"], oi["bucket_count_nonzero"])
        assert ref.has_graphlets(h) == oi["has_graphlets"]
        want = [x.numpy() for x in ref.sample(h, m, k, mode, bo, seed)]
        got = P.sample(m, k, mode, bo, seed)
        for a, b in zip(want, got):
            assert a.shape == b.shape and np.array_equal(a, b)
        ref.destroy_preproc(h)
        P.close()


def test_sample_batch_random_shared_cache(ref):
    """same call sequence on both sides so the process-global LRU (key ignores k) evolves identically."""
    import torch
    rng = random.Random(99)
    cache = oracle.Cache(1000)
    for _ in range(150):
        G = rng.randint(1, 6)
        cols, ptr = [], [0]
        for _g in range(G):
            n = rng.choice([0, 1, 2, 3, 5, 8, 12, 20, 40])
            off = ptr[-1]
            e = [(u + off, v + off) for u in range(n) for v in range(u + 1, n) if rng.random() < rng.choice([0.1, 0.3, 0.7])]
            if n and rng.random() < 0.2:
                e.append((off + rng.randrange(n),) * 2)
            if rng.random() < 0.5:
                e = e + [(v, u) for u, v in e]
            cols += e
            ptr.append(off + n)
        if rng.random() < 0.3:
            rng.shuffle(cols)
        ei = np.array(cols, dtype=np.int64).T.reshape(2, -1)
        ptr = np.array(ptr, dtype=np.int64)
        m, k = rng.choice([1, 2, 7, 33]), rng.randint(1, 7)
        mode, seed = rng.choice(["sample", "graph", "global"]), rng.choice([42, 0, -5, 99991])
        want = [x.numpy() for x in ref.sample_batch(torch.from_numpy(ei), torch.from_numpy(ptr), m, k, mode, seed)]
        got = oracle.sample_batch(ei, ptr, m, k, mode, seed, cache=cache)
        for a, b in zip(want, got):
            assert a.shape == b.shape and np.array_equal(a, b)
