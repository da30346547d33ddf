"""GPU: the device batch pass (SURVEY.md 8(f) N4 for batches of small graphs, csrc/ugs_batch.hip) against the CPU oracle.

The pass replaces the reference's per-graph slicing (src/ugs_sampler_batch_extension.cpp:41-75), LRU key (include/cache.hpp:81-109)
and CSR construction (src/preproc.cpp:32-86) by one pass over edge_index + ptr on the device; the host only replays the LRU on the
keys.  What must hold: the sampler's five output tensors equal the oracle's, call after call, with ONE shared LRU history on each
side (keys ignore k, evictions, reuse) -- through the pass, through the general path, and when calls alternate between them."""
import os
import random
import subprocess
import sys

import numpy as np
import pytest
import torch

import oracle

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(600)]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _graph(rng, n, off):
    p = rng.choice([0.03, 0.1, 0.3, 0.7])
    e = [(u + off, v + off) for u in range(n) for v in range(u + 1, n) if rng.random() < p]
    e = e[:450]                                                              # at most 1000 columns per graph after doubling: the pass's limit
    if rng.random() < 0.6:
        e = e + [(v, u) for u, v in e]                                       # both directions, like PyG
    if rng.random() < 0.3 and n:
        e += [(off + rng.randrange(n),) * 2 for _ in range(3)]               # self loops
    if rng.random() < 0.2:
        e += e[: len(e) // 5]                                                # repeated columns
    return e[:1000]


def _batch(rng, pool):
    """a mini-batch: graphs drawn from `pool` (so that later batches are new combinations of known graphs) plus new ones, with
    empty graphs, graphs smaller than k, cross-graph and out-of-range columns, and optionally interleaved columns"""
    G = rng.randint(1, 12)
    cols, ptr = [], [0]
    for _ in range(G):
        r = rng.random()
        if r < 0.08:
            n, e = 0, []                                                     # empty node range
        elif r < 0.5 and pool:
            n, e0 = pool[rng.randrange(len(pool))]
            e = [(u + ptr[-1], v + ptr[-1]) for u, v in e0]
        else:
            n = rng.choice([1, 2, 3, 5, 9, 18, 39, 70, 130])
            e = _graph(rng, n, ptr[-1])
            pool.append((n, [(u - ptr[-1], v - ptr[-1]) for u, v in e]))
        cols += e
        ptr.append(ptr[-1] + n)
    total = ptr[-1]
    if total and rng.random() < 0.3:
        for _ in range(4):                                                   # columns that belong to no graph
            cols.append((rng.randrange(total), rng.randrange(total)))
        cols += [(-1, 0), (0, total), (total + 5, 1)]
    if rng.random() < 0.35:
        rng.shuffle(cols)                                                    # a graph's columns need not be contiguous
    ei = np.array(cols, dtype=np.int64).T.reshape(2, -1).copy()
    k = rng.choice([1, 2, 3, 4, 5, 6, 8, 12])
    return ei, np.array(ptr, dtype=np.int64), rng.choice([1, 5, 33]), k, rng.choice(["sample", "graph", "global"]), rng.choice([42, 0, -3, 99991])


def _run(calls, env):
    import ugs_sampler
    prev = os.environ.get("UGS_DEVICE_BATCH")
    ugs_sampler.clear_cache()
    cache = oracle.Cache()
    try:
        for it, (ei, ptr, m, k, mode, seed) in enumerate(calls):
            setting = env[it % len(env)]
            if setting is None:
                os.environ.pop("UGS_DEVICE_BATCH", None)
            else:
                os.environ["UGS_DEVICE_BATCH"] = setting
            want = oracle.sample_batch(ei, ptr, m, k, mode, seed, cache)
            got = ugs_sampler.sample_batch(torch.from_numpy(ei), torch.from_numpy(ptr), m, k, mode, seed)
            for name, g, w in zip(("nodes", "edge_index", "edge_ptr", "sample_ptr", "edge_src"), got, want):
                assert np.array_equal(g.numpy(), np.asarray(w)), (it, setting, name, list(np.diff(ptr)), ei.shape[1], m, k, mode, seed)
    finally:
        if prev is None:
            os.environ.pop("UGS_DEVICE_BATCH", None)
        else:
            os.environ["UGS_DEVICE_BATCH"] = prev
        ugs_sampler.clear_cache()
        cache.close()


@pytest.mark.parametrize("env", [("1",), ("0",), (None, "0", "1")])
def test_random_minibatches_through_the_device_pass(env):
    """pass forced wherever it applies, pass off, and calls alternating between default (pass from 2048 columns on), off and forced
    over ONE LRU history"""
    import ugs_sampler
    rng = random.Random(31337)
    pool = []
    calls = [_batch(rng, pool) for _ in range(260)]
    before = ugs_sampler.batch_pass_stats()
    _run(calls, env)
    after = ugs_sampler.batch_pass_stats()
    built = after["device_plans"] - before["device_plans"]
    if env == ("0",):
        assert built == 0
    else:
        assert built > (100 if env == ("1",) else 60), (before, after)       # the pass really served the calls (repeats hit the whole-batch index)


def test_new_combinations_of_known_graphs_and_the_limits_of_the_pass():
    """the trainer's case -- every call a new combination of graphs the LRU knows -- and batches the pass must hand to the general
    path: a graph with more than 1000 columns, more than 2048 vertices, a non-monotone ptr"""
    import ugs_sampler
    import ugs_workloads as wl
    ei, ptr = wl.tu_batch(39, 73, 32)
    G, n_per, cols_per = 32, 39, ei.shape[1] // 32
    rng = np.random.default_rng(5)
    calls = [(ei, ptr, 16, 6, "sample", 42)]
    for _ in range(12):
        perm = rng.permutation(G)
        blocks = [ei[:, g * cols_per:(g + 1) * cols_per] - g * n_per + i * n_per for i, g in enumerate(perm)]
        calls.append((np.ascontiguousarray(np.concatenate(blocks, axis=1)), ptr, 16, 6, "sample", 42))
    s0 = ugs_sampler.batch_pass_stats()
    _run(calls, (None,))                                                       # 4672 columns: the default takes the pass
    s1 = ugs_sampler.batch_pass_stats()
    assert s1["device_plans"] - s0["device_plans"] == 13 and s1["general_path"] == s0["general_path"]
    rr = random.Random(8)
    big_cols = np.array([(u, v) for u in range(60) for v in range(60) if u != v and rr.random() < 0.4], dtype=np.int64).T.reshape(2, -1)   # ~1400 columns
    assert big_cols.shape[1] > 1000
    many = np.array([(i, i + 1) for i in range(2500)], dtype=np.int64).T.reshape(2, -1)                                                   # 2501 vertices
    overl = np.array([(0, 1), (1, 2), (2, 3), (3, 4), (4, 5), (5, 0), (1, 4)], dtype=np.int64).T.reshape(2, -1)
    calls = [(big_cols, np.array([0, 60], dtype=np.int64), 9, 4, "graph", 1), (many, np.array([0, 2501], dtype=np.int64), 9, 3, "global", 2),
             (overl, np.array([2, 6, 0, 5], dtype=np.int64), 9, 3, "global", 3),
             (np.concatenate([big_cols, ei[:, :cols_per] + 60], axis=1), np.array([0, 60, 60 + n_per], dtype=np.int64), 5, 4, "sample", 4)]
    _run(calls, ("1",))
    s2 = ugs_sampler.batch_pass_stats()
    assert s2["device_plans"] == s1["device_plans"] and s2["general_path"] - s1["general_path"] == 4


def test_lru_eviction_through_the_device_pass():
    """UGS_CACHE_SIZE=3 (fixed at first use: subprocess): multi-graph batches whose graphs evict one another, other k's in between"""
    code = r'''
import os, sys, random
os.environ["UGS_CACHE_SIZE"] = "3"
os.environ["UGS_DEVICE_BATCH"] = "1"
sys.path[:0] = [os.path.join(os.getcwd(), p) for p in ("tests", "oracle", "ss-gnn_amd")]
import numpy as np, torch
import oracle, ugs_sampler
rng = random.Random(12)
graphs = []
for n in (6, 8, 10, 12, 14, 9):
    e = [(u, v) for u in range(n) for v in range(u + 1, n) if rng.random() < 0.45]
    graphs.append((n, e + [(v, u) for u, v in e]))
cache = oracle.Cache(3)
for t in range(40):
    picks = [rng.randrange(len(graphs)) for _ in range(rng.randint(1, 4))]
    cols, ptr = [], [0]
    for gi in picks:
        n, e = graphs[gi]
        cols += [(u + ptr[-1], v + ptr[-1]) for u, v in e]
        ptr.append(ptr[-1] + n)
    ei = np.array(cols, dtype=np.int64).T.reshape(2, -1).copy(); ptr = np.array(ptr, dtype=np.int64)
    k = rng.choice([3, 4, 5]); seed = rng.choice([42, 7])
    want = oracle.sample_batch(ei, ptr, 11, k, "sample", seed, cache)
    got = ugs_sampler.sample_batch(torch.from_numpy(ei), torch.from_numpy(ptr), 11, k, "sample", seed)
    assert all(np.array_equal(a.numpy(), np.asarray(b)) for a, b in zip(got, want)), t
assert ugs_sampler.batch_pass_stats()["device_plans"] >= 30
print("OK")
'''
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=ROOT, timeout=200)
    assert out.returncode == 0 and "OK" in out.stdout, out.stderr[-2000:]


def test_many_small_graphs_take_the_two_kernel_variant():
    """G * E above 4 M: the slicing runs as a kernel of its own (thread per column, binary search in ptr, wave-aggregated atomics)
    and the build kernel compacts only the span of its graph -- 900 graphs, shuffled columns, stray columns, against the oracle"""
    rng = random.Random(99)
    cols, ptr = [], [0]
    for g in range(900):
        n = rng.choice([0, 2, 4, 5, 6, 7, 9])
        e = [(u + ptr[-1], v + ptr[-1]) for u in range(n) for v in range(u + 1, n) if rng.random() < 0.5]
        cols += e + [(v, u) for u, v in e]
        ptr.append(ptr[-1] + n)
    cols += [(rng.randrange(ptr[-1]), rng.randrange(ptr[-1])) for _ in range(50)]
    rng.shuffle(cols)
    ei = np.array(cols, dtype=np.int64).T.reshape(2, -1).copy()
    assert 900 * ei.shape[1] > (4 << 20)
    import ugs_sampler
    s0 = ugs_sampler.batch_pass_stats()
    _run([(ei, np.array(ptr, dtype=np.int64), 3, 4, "sample", 42), (ei[:, ::-1].copy(), np.array(ptr, dtype=np.int64), 2, 3, "global", 7)], ("1",))
    assert ugs_sampler.batch_pass_stats()["device_plans"] - s0["device_plans"] == 2


# ---------------------------------------------------------------------------------------------------------------------------
# cold path: graphs the LRU does not know get the REST of their preprocessing on the device too (ugs_bp_roots)
# ---------------------------------------------------------------------------------------------------------------------------
def _fresh_batch(rng, sizes, p_edge, both=True):
    cols, ptr, per = [], [0], []
    for n in sizes:
        e = [(u, v) for u in range(n) for v in range(u + 1, n) if rng.random() < p_edge][:480]
        if both:
            e = e + [(v, u) for u, v in e]
        per.append((n, np.array(e, dtype=np.int64).T.reshape(2, -1)))
        cols += [(u + ptr[-1], v + ptr[-1]) for u, v in e]
        ptr.append(ptr[-1] + n)
    return np.array(cols, dtype=np.int64).T.reshape(2, -1).copy(), np.array(ptr, dtype=np.int64), per


def _expected_roots(n, ei_local, k):
    """what the walk kernels must find for one graph, from the oracle's preprocessing (reference src/preproc.cpp:176-256,
    include/sampler.hpp:44-69, relaxation levels src/sampler.cpp:121-150)"""
    pre = oracle.Preproc(ei_local, n, k)
    d = pre.dump()
    pre.close()
    if (d["bucket_b"] > 0).any():
        return {"level": 0, "prob": d["prob"], "alias": d["alias"], "v_self": d["order"], "v_alias": d["order"][d["alias"]]}
    vi = np.nonzero(d["suffix_deg"] > 0)[0]
    level = 1
    if vi.size == 0:
        level, vi = 2, np.arange(n)
    return {"level": level, "viable_vi": vi.astype(np.int32), "viable_v": d["order"][vi]}


@pytest.mark.parametrize("cold", ["1", "0"])
def test_cold_path_root_records_equal_the_oracle(cold):
    """every graph new to the LRU, pass forced: the records the walk kernels read -- exact prob doubles, alias, both candidate
    vertices, or the viable list of the relaxed levels -- equal the oracle's preprocessing, graph by graph; and so do the samples.
    cold=0 keeps the host preprocessing for unknown graphs (the records then come from the host's alias build)."""
    import ugs_sampler
    rng = random.Random(2024)
    prev = {v: os.environ.get(v) for v in ("UGS_DEVICE_BATCH", "UGS_DEVICE_COLD")}
    os.environ["UGS_DEVICE_BATCH"] = "1"
    os.environ["UGS_DEVICE_COLD"] = cold
    ugs_sampler.clear_cache()
    cache = oracle.Cache()
    try:
        shapes = [([39] * 8, 0.1, 6), ([18] * 12, 0.12, 4), ([5, 3, 9, 2, 7, 30, 64, 1], 0.5, 3), ([12, 12, 12], 0.02, 5),      # sparse: relaxed levels
                  ([200, 150, 310], 0.02, 8), ([700, 40], 0.004, 6), ([1024], 0.003, 5), ([1100, 20], 0.003, 4),                  # 1100 > the device limit of the cold path
                  ([6, 6, 6, 6], 1.0, 6), ([40] * 6, 0.3, 12), ([25] * 4, 0.25, 1), ([25] * 4, 0.25, 2)]
        for it, (sizes, p, k) in enumerate(shapes):
            ei, ptr, per = _fresh_batch(rng, sizes, p, both=it % 3 != 2)
            s0 = ugs_sampler.batch_pass_stats()
            plan = ugs_sampler.Plan.from_batch(torch.from_numpy(ei), torch.from_numpy(ptr), k)
            assert ugs_sampler.batch_pass_stats()["device_plans"] == s0["device_plans"] + 1, (it, sizes)
            for g, (n, el) in enumerate(per):
                if n < k:
                    continue
                got = plan.graph_roots(g, max(n, 1))
                want = _expected_roots(n, el, k)
                assert got["level"] == want["level"] and got["num_nodes"] == n, (it, g, got["level"], want["level"])
                if want["level"] == 0:
                    for name in ("prob", "alias", "v_self", "v_alias"):
                        assert np.array_equal(got[name][:n], want[name]), (it, g, n, name)      # doubles compared exactly
                else:
                    nv = want["viable_vi"].size
                    assert got["num_viable"] == nv
                    assert np.array_equal(got["viable_vi"][:nv], want["viable_vi"]) and np.array_equal(got["viable_v"][:nv], want["viable_v"]), (it, g)
            plan.close()
            for mode, seed in (("sample", 42), ("global", -7)):
                want = oracle.sample_batch(ei, ptr, 9, k, mode, seed, cache)
                got = ugs_sampler.sample_batch(torch.from_numpy(ei), torch.from_numpy(ptr), 9, k, mode, seed)
                for a, b in zip(got, want):
                    assert np.array_equal(a.numpy(), np.asarray(b)), (it, sizes, k, mode)
    finally:
        for v, x in prev.items():
            if x is None:
                os.environ.pop(v, None)
            else:
                os.environ[v] = x
        ugs_sampler.clear_cache()
        cache.close()


def test_all_miss_sequence_with_a_tiny_lru():
    """UGS_CACHE_SIZE=3 (fixed at first use: subprocess), every call a batch of graphs the LRU has never seen or has evicted: stubs
    created and evicted inside one call, repeated new graphs inside one batch, other k's in between, the general path meeting
    stubs (it completes them on the host), and the handle API on a stub's handle -- all five tensors equal the oracle's"""
    code = r'''
import os, sys, random
os.environ["UGS_CACHE_SIZE"] = "3"
sys.path[:0] = [os.path.join(os.getcwd(), p) for p in ("tests", "oracle", "ss-gnn_amd")]
import numpy as np, torch
import oracle, ugs_sampler
rng = random.Random(77)
cache = oracle.Cache(3)
def fresh(n):
    e = [(u, v) for u in range(n) for v in range(u + 1, n) if rng.random() < 0.3]
    return n, e + [(v, u) for u, v in e]
recent = []
for t in range(60):
    picks = []
    for _ in range(rng.randint(1, 6)):
        if recent and rng.random() < 0.3:
            picks.append(recent[rng.randrange(len(recent))])           # a graph of an earlier call: evicted by now, or still a stub
        else:
            picks.append(fresh(rng.choice([4, 7, 12, 20, 39])))
        if rng.random() < 0.2:
            picks.append(picks[-1])                                    # the same new graph twice in one batch
    recent = (recent + picks)[-8:]
    cols, ptr = [], [0]
    for n, e in picks:
        cols += [(u + ptr[-1], v + ptr[-1]) for u, v in e]
        ptr.append(ptr[-1] + n)
    ei = np.array(cols, dtype=np.int64).T.reshape(2, -1).copy(); ptr = np.array(ptr, dtype=np.int64)
    k = rng.choice([3, 4, 6]); seed = rng.choice([42, 7]); mode = rng.choice(["sample", "graph", "global"])
    os.environ["UGS_DEVICE_BATCH"] = "0" if t % 5 == 4 else "1"        # every fifth call through the general path
    want = oracle.sample_batch(ei, ptr, 11, k, mode, seed, cache)
    got = ugs_sampler.sample_batch(torch.from_numpy(ei), torch.from_numpy(ptr), 11, k, mode, seed)
    assert all(np.array_equal(a.numpy(), np.asarray(b)) for a, b in zip(got, want)), (t, k, mode)
    st, ost = ugs_sampler.cache_stats(), cache.stats()
    assert (st["hits"], st["misses"]) == (ost["hits"], ost["misses"]), (t, st, ost)
assert ugs_sampler.batch_pass_stats()["device_plans"] >= 40
print("OK")
'''
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=ROOT, timeout=300)
    assert out.returncode == 0 and "OK" in out.stdout, (out.stdout[-500:], out.stderr[-2000:])


def test_a_stub_completes_itself_on_the_host_when_a_host_path_needs_its_arrays():
    """graphs preprocessed by the device exist on the host without arrays; their handles are ordinary handles of the registry
    (the reference's LRU handles are too), so the handle API must work on them: the stub rebuilds its arrays from the column
    span it kept.  Handles are consecutive: one made right before the batch names the batch's."""
    import ugs_sampler
    rng = random.Random(5150)
    prev = os.environ.get("UGS_DEVICE_BATCH")
    os.environ["UGS_DEVICE_BATCH"] = "1"
    ugs_sampler.clear_cache()
    try:
        sizes, k = [20, 33, 8, 51], 5
        ei, ptr, per = _fresh_batch(rng, sizes, 0.2)
        h0 = ugs_sampler.create_preproc(torch.tensor([[0, 1], [1, 0]]), 2, 2)
        got = ugs_sampler.sample_batch(torch.from_numpy(ei), torch.from_numpy(ptr), 4, k, "sample", 42)
        want = oracle.sample_batch(ei, ptr, 4, k, "sample", 42)
        assert all(np.array_equal(a.numpy(), np.asarray(b)) for a, b in zip(got, want))
        for g, (n, el) in enumerate(per):
            h = h0 + 1 + g
            info = ugs_sampler.get_preproc_info(h)
            pre = oracle.Preproc(el, n, k)
            od, oi = pre.dump(), pre.info()
            assert info["num_nodes"] == n and info["Z"] == oi["Z"] and info["bucket_count_nonzero"] == oi["bucket_count_nonzero"], (g, info, oi)
            d = ugs_sampler.preproc_dump(h)
            for name in ("indptr", "indices", "edge_col", "order", "index_of", "suffix_deg", "bucket_b", "prob", "alias"):
                assert np.array_equal(d[name], od[name]), (g, name)
            a = ugs_sampler.sample(h, 7, k, "flat", 0, 3)
            b = pre.sample(7, k, "flat", 0, 3)
            assert all(np.array_equal(x.numpy(), np.asarray(y)) for x, y in zip(a, b)), g
            pre.close()
        ugs_sampler.destroy_preproc(h0)
    finally:
        if prev is None:
            os.environ.pop("UGS_DEVICE_BATCH", None)
        else:
            os.environ["UGS_DEVICE_BATCH"] = prev
        ugs_sampler.clear_cache()
