"""GPU: seeded random batches against the CPU oracle under every first-tier choice -- many small cases rather than a few
big ones (dense 400-vertex graphs right below the 448-candidate tier's limits, multigraphs, self loops, k up to 32,
degenerate graphs, all numbering modes).  Bit-exact."""
import os
import random

import numpy as np
import pytest
import torch

import oracle

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(600)]


def _case(rng):
    G = rng.randint(1, 4)
    cols, ptr = [], [0]
    for _ in range(G):
        n = rng.choice([5, 12, 30, 70, 150, 400, 460, 1100])
        p = rng.choice([0.02, 0.05, 0.15, 0.4, 0.9])
        if n >= 150:
            p = min(p, 0.4 if n < 1000 else 0.15)
        off = ptr[-1]
        e = [(u + off, v + off) for u in range(n) for v in range(u + 1, n) if rng.random() < p]
        if rng.random() < 0.4:
            e = e + [(v, u) for u, v in e]
        if rng.random() < 0.3:
            e += [(off + rng.randrange(n),) * 2 for _ in range(3)]
        if rng.random() < 0.2:
            e += e[: len(e) // 4]
        if rng.random() < 0.15:                                                  # a hub inside the graph
            e += [(off, off + v) for v in range(1, n)]
        cols += e
        ptr.append(off + n)
    if rng.random() < 0.5:
        rng.shuffle(cols)
    ei = np.array(cols, dtype=np.int64).T.reshape(2, -1).copy()
    k = rng.choice([1, 2, 3, 4, 5, 6, 7, 8, 9, 11, 12, 16, 24, 32])
    return (ei, np.array(ptr, dtype=np.int64), rng.choice([1, 7, 40]), k, rng.choice(["sample", "graph", "global"]),
            rng.choice([42, 0, -7, 123456]))


@pytest.mark.parametrize("tier", [None, "0", "1", "2", "3", "4", "5"])
def test_random_batches_under_every_first_tier(tier, monkeypatch):
    import ugs_sampler
    if tier is None:
        monkeypatch.delenv("UGS_FORCE_TIER", raising=False)
        monkeypatch.setenv("UGS_DEVICE_BATCH", "1")          # this sequence also goes through the device batch pass wherever it applies
    else:
        monkeypatch.setenv("UGS_FORCE_TIER", tier)
    rng = random.Random(2000 + (int(tier) if tier else 7))          # (tier "2" = the 704-candidate tier: dense 400-700-vertex graphs reach its two-pass final)
    ugs_sampler.clear_cache()
    cache = oracle.Cache()        # the reference's LRU lives across calls (its key ignores k): same call history on both sides
    for it in range(300):
        ei, ptr, m, k, mode, seed = _case(rng)
        try:
            want = oracle.sample_batch(ei, ptr, m, k, mode, seed, cache)
        except oracle.OracleError as ex:
            want = ex
        try:
            got = ugs_sampler.sample_batch(torch.from_numpy(ei), torch.from_numpy(ptr), m, k, mode, seed)
        except RuntimeError as ex:
            got = ex
        what = (tier, it, list(np.diff(ptr)), ei.shape[1], m, k, mode, seed)
        if isinstance(want, Exception):
            assert isinstance(got, Exception), what
            continue
        assert not isinstance(got, Exception), (what, got)
        for g, w in zip(got, want):
            assert np.array_equal(g.numpy(), np.asarray(w)), what
    ugs_sampler.clear_cache()
    cache.close()
