"""GPU: the HIP product (through the C ABI, via the ugs_sampler drop-in package) against
  (1) the committed golden fixtures generated from the reference, and
  (2) the CPU oracle on seeded random inputs at sizes the oracle finishes in seconds.
Bit-exact: every output is an integer tensor."""
import os
import random

import numpy as np
import pytest

import scenarios as sc

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(180)]


@pytest.fixture(scope="module")
def product():
    from backends import ProductBackend
    return ProductBackend()


@pytest.fixture(scope="module")
def orc():
    from backends import OracleBackend
    return OracleBackend()


@pytest.mark.parametrize("name", sorted(sc.SCENARIOS))
def test_product_reproduces_golden(name, product):
    calls, expected = sc.load_golden(name)
    got = sc.run_scenario(calls, product)
    sc.check_against_golden(calls, expected, got, f"HIP vs golden {name}")


def _rand_batch(rng, sizes, ps):
    G = rng.randint(1, 6)
    cols, ptr = [], [0]
    for _ in range(G):
        n = rng.choice(sizes)
        off = ptr[-1]
        p = rng.choice(ps)
        e = [(u + off, v + off) for u in range(n) for v in range(u + 1, n) if rng.random() < p]
        if n and rng.random() < 0.2:
            e.append((off + rng.randrange(n),) * 2)
        if rng.random() < 0.5:
            e = e + [(v, u) for u, v in e]
        cols += e
        ptr.append(off + n)
    if rng.random() < 0.3:
        rng.shuffle(cols)
    return np.array(cols, dtype=np.int64).T.reshape(2, -1), np.array(ptr, dtype=np.int64)


def _same(calls, product, orc, what):
    got = sc.run_scenario(calls, product)
    want = sc.run_scenario(calls, orc)
    for i, (g, w) in enumerate(zip(got, want)):
        assert type(g) is type(w), f"{what}: call {i}: {g!r} vs {w!r}"
        if isinstance(g, tuple) and g and isinstance(g[0], str):
            assert w[1] in g[1] or g[1] in w[1], f"{what}: call {i}: error text {g[1]!r} vs {w[1]!r}"
        elif isinstance(g, tuple):
            for j, (a, b) in enumerate(zip(g, w)):
                assert a.shape == b.shape and np.array_equal(a, b), f"{what}: call {i} output {j} differs"
        elif isinstance(g, dict):
            assert g == w


def test_random_batches_vs_oracle(product, orc):
    rng = random.Random(31337)
    calls = []
    for _ in range(60):
        ei, ptr = _rand_batch(rng, [0, 1, 2, 3, 5, 8, 12, 20, 40], [0.1, 0.3, 0.7])
        calls.append(dict(fn="sample_batch", edge_index=ei, ptr=ptr, m=rng.choice([1, 2, 7, 33, 100]), k=rng.randint(1, 8),
                          mode=rng.choice(["sample", "graph", "global"]), seed=rng.choice([42, 0, -5, 99991, 2**31 - 1, -2**31])))
    _same(calls, product, orc, "random batches")


@pytest.mark.parametrize("tier", ["0", "1", "2", "3", "4", "5"])
def test_every_walk_tier_gives_the_same_rows(tier, product, orc, monkeypatch):
    """UGS_FORCE_TIER pins the first walk tier (8 lanes/walk cap 64; 64 lanes cap 448 / 704 / 1024 / 1408 / 2048); graphs whose
    candidate sets outgrow the tier are handed to the next one -- the rows must not depend on any of that."""
    monkeypatch.setenv("UGS_FORCE_TIER", tier)
    rng = random.Random(7 + int(tier))
    calls = []
    for n, p, k, m in [(30, 0.2, 5, 64), (120, 0.3, 6, 80), (300, 0.5, 5, 40), (700, 0.45, 4, 24)]:
        e = [(u, v) for u in range(n) for v in range(u + 1, n) if rng.random() < p]
        ei = np.array(e, dtype=np.int64).T.reshape(2, -1)
        calls.append(dict(fn="sample_batch", edge_index=ei, ptr=np.array([0, n], dtype=np.int64), m=m, k=k, mode="graph", seed=11))
    _same(calls, product, orc, f"tier {tier}")


@pytest.mark.parametrize("env", [{"UGS_NO_PROW": "1"}, {"UGS_PROW_SHIFT": "3"}, {"UGS_PROW_SHIFT": "6", "UGS_PROW_FIRST": "16"},
                                 {"UGS_PROW_SHIFT": "5", "UGS_PROW_FIRST": "32"}])
@pytest.mark.parametrize("tier", ["1", "2", "3"])
def test_row_layouts_of_the_wave_tiers_give_the_same_rows(env, tier, product, orc, monkeypatch):
    """The one-walk-per-wave tiers read a vertex's row from its padded block (header + first entries, the rest of the block and
    of the row on demand) or, without padded rows, through the row pointer: block sizes from 8 to 64 entries, a first fetch
    shorter than the block, and the row-pointer variant must all produce the reference's rows (degrees from 0 to several
    hundred, so that every path -- row inside the first fetch, inside the block, continued in the CSR -- is taken)."""
    monkeypatch.setenv("UGS_FORCE_TIER", tier)
    for key, val in env.items():
        monkeypatch.setenv(key, val)
    rng = random.Random(23)
    calls = []
    for n, p, k, m in [(40, 0.15, 5, 64), (90, 0.35, 6, 80), (260, 0.3, 5, 48), (500, 0.5, 4, 24)]:
        e = [(u, v) for u in range(n) for v in range(u + 1, n) if rng.random() < p]
        e += [(0, 0), (3, 3)] + [(v, u) for u, v in e[: len(e) // 3]]            # self loops, some reversed duplicates
        ei = np.array(e, dtype=np.int64).T.reshape(2, -1)
        calls.append(dict(fn="sample_batch", edge_index=ei, ptr=np.array([0, n + 2], dtype=np.int64), m=m, k=k, mode="sample", seed=5))
    _same(calls, product, orc, f"row layout {env} tier {tier}")


def test_hub_graph_uses_global_memory_tier(product, orc):
    """a hub with 6000 neighbours: candidate sets of thousands of vertices (bucket chain past 2357 / 5087) exceed every
    LDS tier and run in the global-memory workspace tier."""
    rng = random.Random(5)
    n = 6500
    e = [(0, v) for v in range(1, 6001)] + [(rng.randrange(1, n), rng.randrange(1, n)) for _ in range(9000)]
    e = [(u, v) for u, v in e if u != v]
    ei = np.array(e, dtype=np.int64).T.reshape(2, -1)
    calls = [dict(fn="sample_batch", edge_index=ei, ptr=np.array([0, n], dtype=np.int64), m=24, k=4, mode="sample", seed=42)]
    _same(calls, product, orc, "hub graph")


def test_handle_api_vs_oracle(product, orc):
    rng = random.Random(2)
    calls = []
    for slot in range(12):
        n = rng.choice([4, 9, 25, 60])
        e = [(u, v) for u in range(n) for v in range(u + 1, n) if rng.random() < rng.choice([0.1, 0.4])]
        if rng.random() < 0.5:
            e = e + [(v, u) for u, v in e]
        ei = np.array(e, dtype=np.int64).T.reshape(2, -1)
        k = rng.randint(1, 7)
        calls.append(dict(fn="create_preproc", edge_index=ei, num_nodes=n, k=k, slot=slot))
        for em in ("local", "flat", "global"):
            calls.append(dict(fn="sample", slot=slot, m=rng.choice([1, 17, 60]), k=rng.choice([k, max(1, k - 1)]), edge_mode=em,
                              base_offset=rng.choice([0, 1000]), seed=rng.choice([42, -1, 0])))
        calls.append(dict(fn="destroy_preproc", slot=slot))
    _same(calls, product, orc, "handle API")


def test_reference_pytest_assertions_on_gpu_output():
    """the two real pytest tests of the reference (tests/test_sample_batch.py:18-41), on GPU output."""
    import torch
    import ugs_sampler
    ptr = torch.tensor([0, 4, 8], dtype=torch.long)
    edge_index = torch.tensor([[0, 1, 2, 4, 5, 6, 4], [1, 2, 3, 5, 6, 7, 7]], dtype=torch.long)
    nodes_t, edge_index_t, edge_ptr_t, sample_ptr_t, edge_src = ugs_sampler.sample_batch(edge_index, ptr, 2, 3, mode="global")
    for i, (u, v) in enumerate(edge_index_t.t().tolist()):
        iu, iv = edge_index.t()[edge_src[i]].tolist()
        assert (u, v) == (iu, iv) or (u, v) == (iv, iu)
    nodes_t, edge_index_t, edge_ptr_t, sample_ptr_t, edge_src = ugs_sampler.sample_batch(edge_index, ptr, 2, 3, mode="sample")
    ep = edge_ptr_t.tolist()
    for i, (u, v) in enumerate(edge_index_t.t().tolist()):
        g_id = next(j for j in range(len(ep) - 1) if ep[j] <= i < ep[j + 1])
        ug, vg = nodes_t[g_id, u].item(), nodes_t[g_id, v].item()
        iu, iv = edge_index.t()[edge_src[i]].tolist()
        assert (ug, vg) == (iu, iv) or (ug, vg) == (iv, iu)
    assert sample_ptr_t.tolist() == [0, 2, 4] and nodes_t.dtype == torch.int64 and nodes_t.shape == (4, 3)


def test_uniformity_script_statistics_on_gpu_output():
    """reference tests/test_uniformity.py (synthetic 6-ring, k=4, 5000 samples, seed 42) prints: 7 'unique',
    counts 1492/1040/517/510/485/479/477, CV 0.517, POOR.  Identical numbers must come out of the GPU sampler."""
    import json
    import torch
    import ugs_sampler
    from uniformity_stats import script_stats
    ei = torch.tensor([[0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 0], [1, 0, 2, 1, 3, 2, 4, 3, 5, 4, 0, 5]], dtype=torch.long)
    ugs_sampler.clear_cache()
    nodes, edge_index, edge_ptr, _, _ = ugs_sampler.sample_batch(ei, torch.tensor([0, 6]), m_per_graph=5000, k=4, mode="sample")
    st = script_stats(nodes.numpy(), edge_index.numpy(), edge_ptr.numpy(), 4)
    with open(os.path.join(sc.GOLDEN_DIR, "f3_ring_uniformity_stats.json")) as f:
        assert st == json.load(f)["script"]
    assert st["counts"] == [1492, 1040, 517, 510, 485, 479, 477] and st["cv"] == 0.517 and st["verdict"] == "POOR"


def test_device_outputs_and_plan_row_ranges():
    """device= returns the same values on the GPU; Plan.sample_rows over disjoint row ranges concatenates to the
    full result (the basis of multi-GPU sharding)."""
    import torch
    import ugs_sampler
    import ugs_workloads as wl
    ei, ptr = wl.tu_batch(39, 73, 6)
    ei_t, ptr_t = torch.from_numpy(ei), torch.from_numpy(ptr)
    ugs_sampler.clear_cache()
    m, k = 50, 6
    for mode in ("sample", "graph", "global"):
        host = ugs_sampler.sample_batch(ei_t, ptr_t, m, k, mode=mode, seed=9)
        dev = ugs_sampler.sample_batch(ei_t, ptr_t, m, k, mode=mode, seed=9, device="cuda:0")
        assert all(t.is_cuda for t in dev)
        for a, b in zip(host, dev):
            assert torch.equal(a, b.cpu())
        plan = ugs_sampler.Plan.from_batch(ei_t, ptr_t, k)
        rows = 6 * m
        cuts = [0, 1, 77, 150, 151, rows]
        parts = [plan.sample_rows(m, mode=mode, seed=9, row_begin=a, row_count=b - a) for a, b in zip(cuts[:-1], cuts[1:])]
        nodes = torch.cat([p[0] for p in parts]).cpu()
        eidx = torch.cat([p[1] for p in parts], dim=1).cpu()
        esrc = torch.cat([p[3] for p in parts]).cpu()
        counts = torch.cat([p[2][1:] - p[2][:-1] for p in parts]).cpu()
        eptr = torch.cat([torch.zeros(1, dtype=torch.int64), counts.cumsum(0)])
        assert torch.equal(nodes, host[0]) and torch.equal(eidx, host[1]) and torch.equal(eptr, host[2]) and torch.equal(esrc, host[4])
        plan.close()


def test_error_behaviour_matches_reference():
    import torch
    import ugs_sampler
    ei = torch.tensor([[0, 1], [1, 2]], dtype=torch.long)
    ptr = torch.tensor([0, 3], dtype=torch.long)
    with pytest.raises(RuntimeError, match="mode must be one of: 'sample', 'graph', 'global'"):
        ugs_sampler.sample_batch(ei, ptr, 1, 2, mode="nope")
    with pytest.raises(RuntimeError, match="edge_index must be int64"):
        ugs_sampler.sample_batch(ei.to(torch.int32), ptr, 1, 2)
    with pytest.raises(RuntimeError, match="ptr must be int64"):
        ugs_sampler.sample_batch(ei, ptr.to(torch.int32), 1, 2)
    with pytest.raises(RuntimeError, match="edge_index must be on CPU"):
        ugs_sampler.sample_batch(ei.cuda(), ptr, 1, 2)
    with pytest.raises(RuntimeError, match="Invalid preproc handle"):
        ugs_sampler.sample(987654321, 1, 2)
    h = ugs_sampler.create_preproc(torch.zeros((2, 0), dtype=torch.long), 0, 2)
    with pytest.raises(RuntimeError, match="No viable roots available"):
        ugs_sampler.sample(h, 1, 2)
    with pytest.raises(TypeError):
        ugs_sampler.sample_batch(ei, ptr, 1, 2, seed=2**31)
    # keyword call form used by the reference's tools (tools/graphlet_analysis.py:217-224)
    out = ugs_sampler.sample_batch(edge_index=ei, ptr=ptr, m_per_graph=3, k=2, mode="sample", seed=1)
    assert len(out) == 5 and out[0].shape == (3, 2) and out[0].is_pinned()


def test_sharded_path_over_rccl_single_rank():
    """the shipped default row sampler (HIP plan) + the collation over the nccl(=RCCL) backend, world size 1 on the
    one GPU of the box (the multi-rank collation logic itself is covered over gloo on the CPU)."""
    import socket
    import torch
    import torch.distributed as dist
    import ugs_sampler
    import ugs_workloads as wl
    from ugs_sampler import distributed as ud
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        ei, ptr = wl.tu_batch(18, 19, 6)
        ei_t, ptr_t = torch.from_numpy(ei), torch.from_numpy(ptr)
        ugs_sampler.clear_cache()
        for mode, all_ranks in (("sample", True), ("graph", False), ("global", True)):
            want = ugs_sampler.sample_batch(ei_t, ptr_t, 40, 5, mode=mode, seed=3)
            got = ud.sample_batch_sharded(ei_t, ptr_t, 40, 5, mode=mode, seed=3, all_ranks=all_ranks)
            for a, b in zip(got, want):
                assert a.is_cuda and torch.equal(a.cpu(), b)
    finally:
        dist.destroy_process_group()


def _check_size_independent_properties(plan, ei_t, m, k, G):
    """Every emitted edge is a column of the input joining two sampled vertices of its row (edge_src consistency), rows are
    growth-ordered and duplicate-free, edge_ptr is the exclusive scan of per-row counts, repeated calls are identical
    (idempotence), and disjoint row shards concatenate to the unsharded result (digest of digests)."""
    import hashlib
    import torch
    rows = G * m
    full = plan.sample_rows(m, mode="global", seed=42)
    again = plan.sample_rows(m, mode="global", seed=42)
    assert all(torch.equal(a, b) for a, b in zip(full, again))
    nodes, eidx, eptr, esrc = full
    assert int(eptr[0]) == 0 and int(eptr[-1]) == eidx.size(1) == esrc.numel() and bool((eptr[1:] >= eptr[:-1]).all())
    # rows: distinct vertices (complete rows), -1 only as a suffix
    valid = nodes >= 0
    assert bool((valid[:, 1:] <= valid[:, :-1]).all())
    srt = torch.sort(torch.where(valid, nodes, torch.arange(-k, 0, device=nodes.device).expand_as(nodes)), dim=1).values
    assert bool((srt[:, 1:] != srt[:, :-1]).all())
    # edges: endpoints are the column's endpoints (either orientation) and both belong to the row
    cols = ei_t.cuda()[:, esrc]
    same = (cols[0] == eidx[0]) & (cols[1] == eidx[1])
    flip = (cols[0] == eidx[1]) & (cols[1] == eidx[0])
    assert bool((same | flip).all())
    row_of_edge = torch.repeat_interleave(torch.arange(rows, device=nodes.device), eptr[1:] - eptr[:-1])
    assert bool((nodes[row_of_edge] == eidx[0].unsqueeze(1)).any(1).all()) and bool((nodes[row_of_edge] == eidx[1].unsqueeze(1)).any(1).all())
    # incomplete rows carry no edges
    assert bool(((eptr[1:] - eptr[:-1])[~valid.all(1)] == 0).all())
    # shards
    cuts = [0, rows // 3, rows // 3 + 1, rows]
    parts = [plan.sample_rows(m, mode="global", seed=42, row_begin=a, row_count=b - a) for a, b in zip(cuts[:-1], cuts[1:])]
    h = lambda t: hashlib.sha256(t.cpu().numpy().tobytes()).hexdigest()
    assert h(torch.cat([p[0] for p in parts])) == h(nodes) and h(torch.cat([p[1] for p in parts], dim=1)) == h(eidx)
    assert h(torch.cat([p[3] for p in parts])) == h(esrc)
    return full


def test_full_size_properties_er_proxy_and_c4():
    """Size-independent properties at sizes the CPU oracle would take minutes for."""
    import torch
    import ugs_sampler
    import ugs_workloads as wl
    for name in ("er_200000_4000000_300000_8", "c4_qm9_b65536"):
        ei, ptr, m, k = wl.workload(name)
        ei_t, ptr_t = torch.from_numpy(ei), torch.from_numpy(ptr)
        plan = ugs_sampler.Plan.from_batch(ei_t, ptr_t, k)
        _check_size_independent_properties(plan, ei_t, m, k, len(ptr) - 1)
        plan.close()


@pytest.mark.timeout(400)
def test_c5_headline_job_at_full_size():
    """BASELINE.json configs[4] itself -- |V| = 1M, 20M columns, k = 8, batch = 1M rows -- through the plan API: the
    size-independent properties of the whole batch, and ALL FOUR tensors of two row sub-ranges (the first rows and a range in
    the middle of the batch, both numbering modes) bit for bit against the CPU oracle."""
    import torch
    import oracle
    import ugs_sampler
    import ugs_workloads as wl
    ei, ptr, m, k = wl.workload("c5_er_1m")
    assert (m, k, int(ptr[-1])) == (1_000_000, 8, 1_000_000) and ei.shape[1] > 19_990_000
    ei_t, ptr_t = torch.from_numpy(ei), torch.from_numpy(ptr)
    plan = ugs_sampler.Plan.from_batch(ei_t, ptr_t, k)
    assert plan.info()["tier"] == 1                                            # the 448-candidate one-walk-per-wave tier
    full = _check_size_independent_properties(plan, ei_t, m, k, 1)
    nodes, eidx, eptr, esrc = [t.cpu().numpy() for t in full]
    assert nodes.shape == (m, k) and (nodes >= 0).all()                        # every row complete on this graph
    P = oracle.Preproc(ei, 1_000_000, k)
    for lo, n in ((0, 12_000), (777_777, 4_000)):
        w_nodes, w_eidx, w_eptr, w_esrc = P.sample(m, k, "global", 0, 42, lo, lo + n)
        e0, e1 = int(eptr[lo]), int(eptr[lo + n])
        assert np.array_equal(nodes[lo:lo + n], w_nodes) and np.array_equal(eptr[lo:lo + n + 1] - e0, w_eptr)
        assert np.array_equal(eidx[:, e0:e1], w_eidx) and np.array_equal(esrc[e0:e1], w_esrc)
    # the numbering of sample_batch's default mode on a shard of the same job
    lo, n = 400_000, 3_000
    g = [t.cpu().numpy() for t in plan.sample_rows(m, mode="sample", seed=42, row_begin=lo, row_count=n)]
    w = P.sample(m, k, "local", 0, 42, lo, lo + n)
    for a, b in zip(g, w):
        assert np.array_equal(a, b)
    P.close()
    plan.close()


def test_more_edge_cases_vs_oracle(product, orc):
    """k at its maximum (32) on a dense graph, k=1, m=0, zero graphs, isolated vertices, a non-monotone ptr (overlapping
    node ranges: every graph scans every column, like the reference), duplicated columns and self loops everywhere."""
    rng = random.Random(77)
    calls = []
    n = 60
    dense = np.array([(u, v) for u in range(n) for v in range(u + 1, n) if rng.random() < 0.5], dtype=np.int64).T.reshape(2, -1)
    calls.append(dict(fn="sample_batch", edge_index=dense, ptr=np.array([0, n], dtype=np.int64), m=12, k=32, mode="graph", seed=42))
    calls.append(dict(fn="sample_batch", edge_index=dense, ptr=np.array([0, n], dtype=np.int64), m=12, k=1, mode="sample", seed=42))
    calls.append(dict(fn="sample_batch", edge_index=dense, ptr=np.array([0, n], dtype=np.int64), m=0, k=3, mode="sample", seed=42))
    calls.append(dict(fn="sample_batch", edge_index=np.zeros((2, 0), np.int64), ptr=np.array([0], dtype=np.int64), m=4, k=3, mode="sample", seed=42))
    iso = np.array([(0, 1), (1, 2), (5, 6)], dtype=np.int64).T.reshape(2, -1)          # vertices 3, 4, 7, 8 isolated
    calls.append(dict(fn="sample_batch", edge_index=iso, ptr=np.array([0, 9], dtype=np.int64), m=30, k=3, mode="global", seed=5))
    calls.append(dict(fn="sample_batch", edge_index=iso, ptr=np.array([0, 9], dtype=np.int64), m=30, k=2, mode="global", seed=5))
    multi = np.array([(0, 1), (0, 1), (1, 0), (1, 1), (1, 2), (2, 2), (2, 3), (3, 0), (3, 0)], dtype=np.int64).T.reshape(2, -1)
    calls.append(dict(fn="sample_batch", edge_index=multi, ptr=np.array([0, 4], dtype=np.int64), m=50, k=3, mode="sample", seed=9))
    overl = np.array([(0, 1), (1, 2), (2, 3), (3, 4), (4, 5), (5, 0), (1, 4)], dtype=np.int64).T.reshape(2, -1)
    calls.append(dict(fn="sample_batch", edge_index=overl, ptr=np.array([2, 6, 0, 5], dtype=np.int64), m=9, k=3, mode="global", seed=3))
    _same(calls, product, orc, "edge cases")


@pytest.mark.parametrize("packed", [True, False])
def test_drop_in_call_with_the_step_run_in_begin(packed, product, orc, monkeypatch):
    """Batches of small graphs: the drop-in call runs walk + fill as one step while it waits for the total (the outputs staged for
    that total, ugs_fill_scan's packed form) and only copies out afterwards -- against the oracle for every mode, a batch whose
    repeated columns push the total past the staging's bound (the kernel writes nothing, the ordinary fill runs), rows that end
    inside a tile, k = 2, device outputs; UGS_NO_PACKED_STEP=1 is the two-phase form."""
    import torch
    import ugs_sampler
    import ugs_workloads as wl
    if not packed:
        monkeypatch.setenv("UGS_NO_PACKED_STEP", "1")
    calls = []
    for (ei, ptr), k, m in [(wl.tu_batch(39, 73, 32), 6, 64), (wl.tu_batch(18, 20, 9), 4, 77), (wl.tu_batch(25, 40, 3), 2, 11)]:
        for mode in ("sample", "graph", "global"):
            calls.append(dict(fn="sample_batch", edge_index=ei, ptr=ptr, m=m, k=k, mode=mode, seed=7))
    ei, ptr = wl.tu_batch(12, 30, 8)
    rep = np.ascontiguousarray(np.concatenate([ei] * 5, axis=1))                   # every column five times: ~5 x the entries of a simple graph
    calls.append(dict(fn="sample_batch", edge_index=rep, ptr=ptr, m=64, k=4, mode="sample", seed=3))
    _same(calls, product, orc, "packed step" if packed else "two-phase step")
    import oracle
    want = oracle.sample_batch(rep, ptr, 64, 4, "sample", 3)
    assert int(np.asarray(want[2])[-1]) > 8 * 64 * 2 * 4 * 3, "the repeated-column batch must exceed the staging's bound"
    # device outputs
    dev = torch.device("cuda:0")
    (ei, ptr), k, m = (wl.tu_batch(39, 73, 32), 6, 64)
    got = ugs_sampler.sample_batch(torch.from_numpy(ei), torch.from_numpy(ptr), m, k, mode="graph", seed=7, device=dev)
    want = oracle.sample_batch(ei, ptr, m, k, "graph", 7)
    for g, w in zip(got, want):
        assert g.device.type == "cuda" and np.array_equal(g.cpu().numpy(), np.asarray(w))


def test_lru_eviction_with_small_cache_matches_oracle():
    """UGS_CACHE_SIZE=2 (read once per process, like the reference): evictions and re-creations with another k must follow
    the reference's LRU exactly.  Runs in a subprocess because the capacity is fixed at first use."""
    import subprocess
    import sys
    code = r'''
import os, sys, random
os.environ["UGS_CACHE_SIZE"] = "2"
sys.path[:0] = [os.path.join(os.getcwd(), p) for p in ("tests", "oracle", "ss-gnn_amd")]
import numpy as np
import scenarios as sc
from backends import ProductBackend, OracleBackend
rng = random.Random(4)
graphs = []
for n in (7, 9, 11, 13):
    e = [(u, v) for u in range(n) for v in range(u + 1, n) if rng.random() < 0.4]
    graphs.append((n, np.array(e + [(v, u) for u, v in e], dtype=np.int64).T.reshape(2, -1)))
calls = []
for t in range(24):
    n, ei = graphs[rng.randrange(4)]
    calls.append(dict(fn="sample_batch", edge_index=ei, ptr=np.array([0, n], dtype=np.int64), m=20, k=rng.choice([3, 4, 5]), mode="sample", seed=42))
got = sc.run_scenario(calls, ProductBackend())
want = sc.run_scenario(calls, OracleBackend(cache_capacity=2))
assert all(all(np.array_equal(a, b) for a, b in zip(g, w)) for g, w in zip(got, want))
print("OK")
'''
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))), timeout=150)
    assert out.returncode == 0 and "OK" in out.stdout, out.stderr[-2000:]


@pytest.mark.parametrize("tier", ["1", "2", "3", "4", "5"])
def test_edges_staged_by_the_walk_and_the_leftover_rows(tier, product, orc, monkeypatch):
    """64-lane tiers: the walk stages up to 32 induced-edge hits per row and the fill kernel expands them; denser rows go
    to the row-reading fill kernel through a list.  Dense graphs with large k (hundreds of induced edges), multigraphs with
    repeated columns and self loops, every numbering mode."""
    monkeypatch.setenv("UGS_FORCE_TIER", tier)
    rng = random.Random(100 + int(tier))
    calls = []
    for n, p, k, m, mode in [(24, 0.9, 12, 60, "sample"), (40, 0.5, 9, 80, "graph"), (60, 0.25, 8, 100, "global"), (18, 1.0, 10, 40, "sample")]:
        e = [(u, v) for u in range(n) for v in range(u + 1, n) if rng.random() < p]
        e += [(v, u) for u, v in e[: len(e) // 2]]                        # both directions of half the edges: repeated pairs
        e += [(rng.randrange(n),) * 2 for _ in range(4)]                  # self loops
        e += e[:7]                                                        # repeated columns
        rng.shuffle(e)
        ei = np.array(e, dtype=np.int64).T.reshape(2, -1).copy()
        calls.append(dict(fn="sample_batch", edge_index=ei, ptr=np.array([0, n], dtype=np.int64), m=m, k=k, mode=mode, seed=21))
    # rows around the 32-hit limit inside ONE call (sparse and dense graphs in a batch)
    cols, ptr = [], [0]
    for n, p in [(30, 0.15), (14, 1.0), (30, 0.5), (12, 0.8)]:
        off = ptr[-1]
        cols += [(u + off, v + off) for u in range(n) for v in range(u + 1, n) if rng.random() < p]
        ptr.append(off + n)
    ei = np.array(cols, dtype=np.int64).T.reshape(2, -1).copy()
    for mode in ("sample", "graph", "global"):
        calls.append(dict(fn="sample_batch", edge_index=ei, ptr=np.array(ptr, dtype=np.int64), m=50, k=9, mode=mode, seed=5))
    _same(calls, product, orc, f"staged edges, tier {tier}")


def test_fill_after_another_walk_reads_the_rows_again(monkeypatch):
    """The staging belongs to the LAST walk of a plan: walk A, walk B, fill A, fill B must still give A's and B's edges."""
    import torch
    import ugs_sampler
    monkeypatch.setenv("UGS_FORCE_TIER", "1")
    rng = random.Random(3)
    n, k, m = 50, 6, 200
    e = [(u, v) for u in range(n) for v in range(u + 1, n) if rng.random() < 0.2]
    ei_t = torch.tensor(e, dtype=torch.long).t().contiguous()
    ptr_t = torch.tensor([0, n], dtype=torch.long)
    ugs_sampler.clear_cache()
    want_a = ugs_sampler.sample_batch(ei_t, ptr_t, m, k, mode="sample", seed=1)
    want_b = ugs_sampler.sample_batch(ei_t, ptr_t, m, k, mode="sample", seed=2)
    plan = ugs_sampler.Plan.from_batch(ei_t, ptr_t, k)
    na, pa, ta = plan.walk(m, "sample", 1)
    nb, pb, tb = plan.walk(m, "sample", 2)
    ea, sa = plan.fill(m, na, pa, ta, "sample")          # staging now holds B's rows: A is filled from the adjacency rows
    eb, sb = plan.fill(m, nb, pb, tb, "sample")          # B: expanded from the staging
    for got, want in ((ea, want_a[1]), (sa, want_a[4]), (eb, want_b[1]), (sb, want_b[4]), (na, want_a[0]), (nb, want_b[0])):
        assert torch.equal(got.cpu(), want)
    # same nodes buffer reused by a later walk, partial row range
    nodes = torch.empty((m, k), dtype=torch.int64, device="cuda")
    eptr = torch.empty((m + 1,), dtype=torch.int64, device="cuda")
    plan.walk(m, "sample", 1, out=(nodes, eptr))
    n2, p2, t2 = plan.walk(m, "sample", 2, row_begin=10, row_count=50, out=(nodes[:50], eptr[:51]))
    e2, s2 = plan.fill(m, n2, p2, t2, "sample", row_begin=10)
    lo, hi = int(want_b[2][10]), int(want_b[2][60])
    assert torch.equal(e2.cpu(), want_b[1][:, lo:hi]) and torch.equal(s2.cpu(), want_b[4][lo:hi])
    plan.close()
    ugs_sampler.clear_cache()


def test_step_captured_as_hip_graph_equals_the_launched_step(monkeypatch):
    """Plan.graph_step: walk tiers + scan + fill captured once and replayed with new seeds give exactly what the separately
    launched step gives -- TU-shaped batch (8-lane tier, row-reading fill) and an ER graph (64-lane tier, staged fill),
    every mode, a row sub-range; a later, larger ordinary call on the same plan must not disturb the captured scratch."""
    import torch
    import ugs_sampler
    import ugs_workloads as wl
    ugs_sampler.clear_cache()
    ei, ptr = wl.tu_batch(39, 73, 8)
    er, erptr = wl.er_graph(3000, 60000, seed=2)
    for (e, p, m, k) in ((ei, ptr, 64, 6), (er, erptr, 500, 6)):
        e_t, p_t = torch.from_numpy(e), torch.from_numpy(p)
        plan = ugs_sampler.Plan.from_batch(e_t, p_t, k)
        rows = (len(p) - 1) * m
        for mode in ("sample", "graph", "global"):
            for (rb, rc) in ((0, rows), (7, rows // 2)):
                step = plan.graph_step(m, mode, row_begin=rb, row_count=rc)
                for seed in (42, 0, -5, 99):
                    got = step.launch(seed).result()
                    want = plan.sample_rows(m, mode=mode, seed=seed, row_begin=rb, row_count=rc)
                    for g, w in zip(got, want):
                        assert torch.equal(g, w), (mode, rb, rc, seed)
                    if seed == 0:
                        plan.sample_rows(2 * m, mode=mode, seed=1)           # regrows the plan's own scratch
                step.close()
        plan.close()
    ugs_sampler.clear_cache()


def test_staging_switched_off_by_its_memory_bound(product, orc, monkeypatch):
    """UGS_STAGE_MAX_MB bounds the staging scratch (512 B per row); a call above the bound takes the row-reading fill kernel
    for every row of the 64-lane tiers -- same output."""
    monkeypatch.setenv("UGS_FORCE_TIER", "1")
    monkeypatch.setenv("UGS_STAGE_MAX_MB", "0")
    rng = random.Random(77)
    n = 90
    e = [(u, v) for u in range(n) for v in range(u + 1, n) if rng.random() < 0.15]
    ei = np.array(e + [(v, u) for u, v in e[:40]], dtype=np.int64).T.reshape(2, -1).copy()
    calls = [dict(fn="sample_batch", edge_index=ei, ptr=np.array([0, n], dtype=np.int64), m=300, k=7, mode=mode, seed=3)
             for mode in ("sample", "graph", "global")]
    _same(calls, product, orc, "staging off")


@pytest.mark.parametrize("tier", [None, "1", "2", "3"])
def test_large_launches_share_work_through_a_counter(tier, product, orc, monkeypatch):
    """Launches with many more walks than resident groups hand the walks out dynamically (chunks of 4 rows per atomicAdd);
    which wave samples a row must not matter: same rows as the oracle, and as the static split (UGS_STATIC_SPLIT)."""
    import torch
    import ugs_sampler
    import ugs_workloads as wl
    if tier is None:
        monkeypatch.delenv("UGS_FORCE_TIER", raising=False)
    else:
        monkeypatch.setenv("UGS_FORCE_TIER", tier)
    ei, ptr = wl.tu_batch(39, 73, 8)
    m = 40000 if tier is None else 6000                  # 320 000 rows in the 8-lane tier, 48 000 in a 64-lane tier
    calls = [dict(fn="sample_batch", edge_index=ei, ptr=ptr, m=m, k=4, mode="sample", seed=17)]
    _same(calls, product, orc, f"dynamic split, tier {tier}")
    dyn = ugs_sampler.sample_batch(torch.from_numpy(ei), torch.from_numpy(ptr), m, 4, "sample", 18)
    monkeypatch.setenv("UGS_STATIC_SPLIT", "1")
    sta = ugs_sampler.sample_batch(torch.from_numpy(ei), torch.from_numpy(ptr), m, 4, "sample", 18)
    for a, b in zip(dyn, sta):
        assert torch.equal(a, b)


def test_dynamic_split_with_rows_handed_on_to_the_next_tier():
    """ER degree 40, k = 12: the 448-candidate tier starts, a few per cent of the walks outgrow it and are redone by the
    1024-candidate tier from its overflow list -- both launches large enough to use the shared work counter."""
    import torch
    import ugs_sampler
    import oracle
    import ugs_workloads as wl
    os.environ.pop("UGS_FORCE_TIER", None)
    n, k, m = 60000, 12, 40000
    ei, ptr = wl.er_graph(n, 1_200_000, seed=11)
    ugs_sampler.clear_cache()
    plan = ugs_sampler.Plan.from_batch(torch.from_numpy(ei), torch.from_numpy(ptr), k)
    nodes, eptr, total = plan.walk(m, "sample", 5)
    handed_on = plan.last_launch()["overflow_rows"]
    eidx, esrc = plan.fill(m, nodes, eptr, total, "sample")
    assert plan.info()["tier"] == 1 and handed_on > 0, (plan.info(), handed_on)
    P = oracle.Preproc(ei, n, k)
    want = P.sample(m, k, "local", 0, 5)
    for g, w in zip((nodes, eidx, eptr, esrc), want):
        assert np.array_equal(g.cpu().numpy(), np.asarray(w))
    P.close()
    plan.close()
    ugs_sampler.clear_cache()


def test_encoder_inputs_equal_what_the_consumer_derives():
    """N1 epilogue: Plan.encoder_inputs (fill numbering row*k + local index, clamped nodes, mask, batch vector) against the torch
    operations of the reference's encoder (src/gps/gps/models/ss_gnn.py:441-468) applied to the ordinary output -- including rows
    of a degenerate graph (all -1) and a row sub-range."""
    import torch
    import ugs_sampler
    import ugs_workloads as wl
    ei, ptr = wl.tu_batch(18, 20, 6)
    ptr = np.concatenate([ptr, [ptr[-1] + 2]])                                 # a last graph with 2 < k vertices: rows of -1
    ei_t, ptr_t = torch.from_numpy(ei), torch.from_numpy(ptr)
    m, k = 40, 4
    plan = ugs_sampler.Plan.from_batch(ei_t, ptr_t, k)
    for row_begin, row_count in ((0, None), (37, 150)):
        nodes, eidx, eptr, esrc = plan.sample_rows(m, "sample", 9, row_begin, row_count)
        B = nodes.size(0)
        dev = nodes.device
        stacked = nodes.flatten()
        want_edge = torch.repeat_interleave(torch.arange(0, B, device=dev), eptr[1:] - eptr[:-1]) * k + eidx
        got = plan.encoder_inputs(m, 9, row_begin, row_count)
        assert torch.equal(got[0], stacked.clamp(min=0)) and torch.equal(got[1], stacked >= 0)
        assert torch.equal(got[2], want_edge) and torch.equal(got[3], esrc)
        assert torch.equal(got[4], torch.repeat_interleave(torch.arange(0, B, device=dev), k))
    assert not bool(plan.encoder_inputs(m, 9)[1].all())                        # the degenerate graph's rows are masked out
    plan.close()


def test_presample_cache_assembles_batches_like_the_reference_trainer():
    """ugs_sampler.presample.PresampleCache.load against a line-by-line numpy restatement of the reference's host loop
    (gps/experiment.py:936-993) over the same cached per-graph results: nodes + ptr[g] (placeholder rows included), local edge
    ids, edge_ptr / sample_ptr accumulation, edge_src + the number of batch columns owned by earlier graphs; a graph that was
    never presampled takes the reference's placeholder path."""
    import torch
    import ugs_sampler
    import ugs_workloads as wl
    from ugs_sampler.presample import PresampleCache
    m, k = 12, 4
    graphs = [wl.tu_graph(n, e, 100 + i) for i, (n, e) in enumerate([(18, 20), (11, 14), (25, 40), (3, 2), (18, 19), (30, 45)])]
    sizes = [18, 11, 25, 3, 18, 30]
    cache = PresampleCache(m, k, "cuda:0")
    host = {}
    for i, (ei, n) in enumerate(zip(graphs, sizes)):
        if i == 4:
            continue                                                        # never presampled: placeholder path
        t = torch.from_numpy(ei)
        cache.add(i, t, n, seed=42 + i)
        host[i] = [x.numpy() for x in ugs_sampler.sample_batch(t, torch.tensor([0, n]), m, k, mode="sample", seed=42 + i)]
    for order in ([0, 1, 2, 3, 4, 5], [5, 2, 2, 4, 0], [3]):
        ptr = np.cumsum([0] + [sizes[i] for i in order])
        cols = np.concatenate([graphs[i] + ptr[j] for j, i in enumerate(order)], axis=1)
        got = [t.cpu().numpy() for t in cache.load(torch.tensor(order), torch.from_numpy(ptr), torch.from_numpy(cols))]
        # the reference's loop
        off = [0]
        for g in range(len(order) - 1):
            off.append(off[-1] + int(((cols[0] >= ptr[g]) & (cols[0] < ptr[g + 1])).sum()))
        nodes, edges, esrc, eptr, sptr, ce, cs = [], [], [], [0], [0], 0, 0
        for g, i in enumerate(order):
            if i in host:
                n_g, e_g, p_g, _, s_g = host[i]
            else:
                n_g, e_g, p_g, s_g = np.full((m, k), -1, np.int64), np.zeros((2, 0), np.int64), np.zeros(m + 1, np.int64), np.zeros(0, np.int64)
            nodes.append(n_g + ptr[g]); edges.append(e_g); esrc.append(s_g + off[g])
            eptr += [ce + int(p_g[r + 1]) for r in range(n_g.shape[0])]
            ce += e_g.shape[1]; cs += n_g.shape[0]; sptr.append(cs)
        want = [np.concatenate(nodes), np.concatenate(edges, axis=1), np.array(eptr), np.array(sptr), np.concatenate(esrc)]
        for a, b in zip(got, want):
            assert np.array_equal(a, b), order


@pytest.mark.parametrize("tier", [None, "1", "2", "3", "4", "5"])
def test_random_large_graphs_through_every_tier(tier, product, orc, monkeypatch):
    """Mid-size random multigraphs (2 000 - 20 000 vertices, degree 6 - 300, columns in one or both directions, k up to 12):
    candidate counts from a handful to more than a thousand, i.e. every variant of the order stages -- member masks of the
    stages of <= 64 and <= 128 elements, the bucket-table finals with 2 ... 17 elements per lane, the materialised 257- and
    541-element stages -- in every LDS tier (forced) and with the tier chosen by the host; all five outputs against the oracle."""
    if tier is None:
        monkeypatch.delenv("UGS_FORCE_TIER", raising=False)
    else:
        monkeypatch.setenv("UGS_FORCE_TIER", tier)
    rng = random.Random(2024 + (int(tier) if tier else 0))
    calls = []
    for _ in range(8):
        nv = rng.choice([2000, 5000, 20000])
        deg = rng.choice([6, 20, 40, 80, 160, 300])
        g = np.random.default_rng(rng.randrange(1 << 30))
        ei = g.integers(0, nv, size=(2, nv * deg // 2), dtype=np.int64)
        if rng.random() < 0.3:
            ei = np.concatenate([ei, ei[::-1]], axis=1)
        calls.append(dict(fn="sample_batch", edge_index=ei, ptr=np.array([0, nv], dtype=np.int64), m=rng.choice([64, 700, 2000]),
                          k=rng.choice([3, 5, 8, 8, 10, 12]), mode=rng.choice(["sample", "graph", "global"]), seed=rng.choice([42, 0, 123456789])))
    _same(calls, product, orc, f"large random graphs, tier {tier}")


def test_whole_batch_index_on_a_batch_hashed_in_chunks(product, orc):
    """A batch of more than 4 M columns is hashed in 8 MB chunks by helper threads (ugs_host.cpp: hash_array).  The same batch
    again must give the same result (served through the index), and ONE changed column in the middle of a chunk, or a changed
    last column, must be noticed -- a stale plan would reproduce the first result instead of the oracle's for the new batch."""
    g = np.random.default_rng(404)
    nv, cols = 60000, (1 << 22) + 12345
    ei = g.integers(0, nv, size=(2, cols), dtype=np.int64)
    ptr = np.array([0, nv], dtype=np.int64)
    base = dict(fn="sample_batch", ptr=ptr, m=300, k=6, mode="sample", seed=42)
    ei2 = ei.copy(); ei2[1, (1 << 21) + 777] = (ei2[1, (1 << 21) + 777] + 1) % nv
    ei3 = ei.copy(); ei3[0, cols - 1] = (ei3[0, cols - 1] + 1) % nv
    calls = [dict(base, edge_index=ei), dict(base, edge_index=ei), dict(base, edge_index=ei2), dict(base, edge_index=ei3), dict(base, edge_index=ei)]
    _same(calls, product, orc, "chunk-hashed batch")


@pytest.mark.parametrize("fused", [True, False])
def test_tier_s_with_16_lanes_per_walk_gives_the_oracle_rows(fused, monkeypatch):
    """Batches of small graphs whose walks are all resident at once run with 16 lanes per walk (ugs_walk_lds<16,64> / <16,32>; 8 lanes
    otherwise and with UGS_NO_WIDE_TIER=1): same rows as the oracle and as the 8-lane form, with and without the scan folded into the
    fill (the 16-lane form leaves the sums of 8 rows through two waves), row ranges that end inside a block of 16 rows."""
    import torch
    import oracle
    import ugs_sampler
    import ugs_workloads as wl
    if not fused:
        monkeypatch.setenv("UGS_NO_FUSED_SCAN", "1")
    torch.cuda.set_device(0)
    ugs_sampler.clear_cache()
    # (the last two: graphs of at most 33 vertices -- the 32-candidate form, ugs_walk_lds<16,32> / <8,32>)
    for (ei, ptr), k, m in [(wl.tu_batch(39, 73, 32), 6, 256), (wl.tu_batch(45, 90, 5), 7, 37), (wl.tu_batch(60, 120, 8), 5, 500),
                            (wl.tu_batch(18, 20, 32), 4, 32), (wl.tu_batch(18, 19, 9), 5, 300)]:
        G = len(ptr) - 1
        rows_all = G * m
        want = oracle.sample_batch(ei, ptr, m, k, "sample", 11)
        outs = {}
        for form in ("wide", "narrow"):
            if form == "narrow":
                monkeypatch.setenv("UGS_NO_WIDE_TIER", "1")
            else:
                monkeypatch.delenv("UGS_NO_WIDE_TIER", raising=False)
            plan = ugs_sampler.Plan.from_batch(torch.from_numpy(ei), torch.from_numpy(ptr), k)
            got = []
            for rb, rc in [(0, rows_all), (3, rows_all - 9), (rows_all - 21, 21), (0, 15), (0, 17)]:
                nodes, eptr, eidx, esrc = plan.step(m, "sample", 11, rb, rc, edge_capacity=int(np.asarray(want[2])[-1]) + 8)
                total = int(eptr[-1].item())
                kern = plan.last_launch()["kernel"]
                assert kern.startswith("ugs_walk_lds<16," if form == "wide" and rc <= 256 * 3 * 16 else "ugs_walk_lds<8,"), (kern, form, rc)
                assert np.array_equal(nodes.cpu().numpy(), np.asarray(want[0])[rb:rb + rc])
                lo, hi = int(np.asarray(want[2])[rb]), int(np.asarray(want[2])[rb + rc])
                assert total == hi - lo
                assert np.array_equal(eidx[:, :total].cpu().numpy(), np.asarray(want[1])[:, lo:hi])
                assert np.array_equal(esrc[:total].cpu().numpy(), np.asarray(want[4])[lo:hi])
                got.append((nodes.cpu(), eptr.cpu(), eidx[:, :total].cpu()))
            outs[form] = got
            plan.close()
        for a, b in zip(outs["wide"], outs["narrow"]):
            assert all(torch.equal(x, y) for x, y in zip(a, b))


@pytest.mark.parametrize("fused", [True, False])
def test_step_in_one_call_equals_walk_then_fill(fused, monkeypatch):
    """Plan.step (walk + fill with the scan folded into the fill kernel for batches of small graphs: tiles of 32 rows in ticket
    order, decoupled look-back) against the oracle and against walk-then-fill: many consecutive steps on one plan (the ticket
    counter and the launch epoch move on, no memset in between), row ranges that are no multiple of a tile, a single row, more
    tiles than one look-back window, every mode; a one-walk-per-wave plan takes the three-launch form behind the same call."""
    import torch
    import oracle
    import ugs_sampler
    import ugs_workloads as wl
    if not fused:
        monkeypatch.setenv("UGS_NO_FUSED_SCAN", "1")
    torch.cuda.set_device(0)
    ugs_sampler.clear_cache()
    cases = [(wl.tu_batch(39, 73, 32), 6, 256), (wl.tu_batch(18, 19, 32), 5, 700), (wl.tu_batch(18, 20, 7), 4, 33)]
    for (ei, ptr), k, m in cases:
        G = len(ptr) - 1
        plan = ugs_sampler.Plan.from_batch(torch.from_numpy(ei), torch.from_numpy(ptr), k)
        rows_all = G * m
        ranges = [(0, rows_all), (5, rows_all - 5), (rows_all // 3, 1), (17, min(rows_all - 17, 2049)), (0, 31), (0, 32), (0, 33)]
        for it, (rb, rc) in enumerate(ranges):
            mode, seed = ("sample", "graph", "global")[it % 3], 42 + it
            n2, p2, tot = plan.walk(m, mode, seed, rb, rc)
            e2, s2 = plan.fill(m, n2, p2, tot, mode, rb)
            nodes, eptr, eidx, esrc = plan.step(m, mode, seed, rb, rc, edge_capacity=tot + 7)
            total = int(eptr[-1].item())
            assert total == tot and torch.equal(nodes, n2) and torch.equal(eptr, p2)
            assert torch.equal(eidx[:, :total], e2) and torch.equal(esrc[:total], s2), (k, m, rb, rc, mode)
            if it < 2:
                want = oracle.sample_batch(ei, ptr, m, k, mode, seed)
                assert np.array_equal(nodes.cpu().numpy(), np.asarray(want[0])[rb:rb + rc])
                lo, hi = int(np.asarray(want[2])[rb]), int(np.asarray(want[2])[rb + rc])
                assert np.array_equal(eidx[:, :total].cpu().numpy(), np.asarray(want[1])[:, lo:hi])
                assert np.array_equal(esrc[:total].cpu().numpy(), np.asarray(want[4])[lo:hi])
        if fused:
            assert plan.last_launch()["kernel"].startswith(("ugs_walk_lds<8", "ugs_walk_lds<16"))
        plan.close()
    # capacity below the total: nothing is written past the buffers, edge_ptr still carries the true total
    (ei, ptr), k, m = cases[0]
    plan = ugs_sampler.Plan.from_batch(torch.from_numpy(ei), torch.from_numpy(ptr), k)
    n2, p2, tot = plan.walk(m, "sample", 1)
    e2, s2 = plan.fill(m, n2, p2, tot, "sample")
    cap = tot // 2                                              # ld is the row stride AND the capacity: contiguous buffers of exactly `cap` entries
    guard = torch.full((2 * cap + 128,), -7, dtype=torch.int64, device="cuda")
    gsrc = torch.full((cap + 64,), -7, dtype=torch.int64, device="cuda")
    out = plan.step(m, "sample", 1, out=(torch.empty_like(n2), torch.empty_like(p2), guard[:2 * cap].view(2, cap), gsrc[:cap]))
    assert int(out[1][-1].item()) == tot and torch.equal(out[2], e2[:, :cap]) and torch.equal(out[3], s2[:cap])
    assert bool((guard[2 * cap:] == -7).all()) and bool((gsrc[cap:] == -7).all())
    plan.close()
    # a large graph (one walk per wave, staged edges): the same call, three launches
    ei, ptr = wl.er_graph(3000, 60000, seed=3)
    plan = ugs_sampler.Plan.from_batch(torch.from_numpy(ei), torch.from_numpy(ptr), 6)
    n2, p2, tot = plan.walk(500, "global", 9)
    e2, s2 = plan.fill(500, n2, p2, tot, "global")
    nodes, eptr, eidx, esrc = plan.step(500, "global", 9, edge_capacity=tot)
    assert torch.equal(nodes, n2) and torch.equal(eptr, p2) and torch.equal(eidx, e2) and torch.equal(esrc, s2)
    plan.close()


def _er(rng, n, deg):
    e = rng.integers(0, n, size=(2, n * deg // 2), dtype=np.int64)
    return e[:, e[0] != e[1]]


@pytest.mark.parametrize("shape", ["small_graphs", "er_degree_20", "er_degree_90"])
def test_streamed_call_equals_the_two_phase_call_and_the_oracle(shape, monkeypatch):
    """ugs_sample_batch_stream (rows in chunks, copy-out beside the walks, row 1 of edge_index last) against the two-phase call of
    the product and against the CPU oracle: all five tensors, every mode, chunks that cut through graphs, a last chunk of one row,
    one chunk for everything, the estimate exactly at the total; a call that outgrows its estimate falls back and says so."""
    import ctypes as C
    import torch
    import oracle
    import ugs_sampler
    from ugs_sampler._lib import lib, UGS_E_CAPACITY
    rng = np.random.default_rng(5)
    if shape == "small_graphs":
        pr = random.Random(9)
        ei, ptr = _rand_batch(pr, [5, 9, 17, 30], [0.3, 0.6])
        while len(ptr) - 1 < 4:
            ei, ptr = _rand_batch(pr, [5, 9, 17, 30], [0.3, 0.6])
        m, k = 301, 4
    elif shape == "er_degree_20":
        ei, ptr, m, k = _er(rng, 3000, 20), np.array([0, 3000], dtype=np.int64), 2501, 8     # one walk per wave, staged edges
    else:
        ei, ptr, m, k = _er(rng, 1500, 90), np.array([0, 1500], dtype=np.int64), 900, 6
    ei = np.ascontiguousarray(ei)                                   # (_rand_batch hands out a transposed view; the direct calls below pass raw pointers)
    ei_t, ptr_t = torch.from_numpy(ei), torch.from_numpy(ptr)
    monkeypatch.setattr(ugs_sampler, "_STREAM_MIN_ROWS", 1)
    ugs_sampler._stream_totals.clear()
    cache = oracle.Cache(1000)
    B = (len(ptr) - 1) * m
    for mode, chunk in (("sample", 97), ("graph", B - 1), ("global", B + 5), ("sample", 1000)):
        monkeypatch.setenv("UGS_NO_STREAMED_CALL", "1")
        two_phase = ugs_sampler.sample_batch(ei_t, ptr_t, m, k, mode=mode, seed=7)
        monkeypatch.delenv("UGS_NO_STREAMED_CALL")
        monkeypatch.setenv("UGS_STREAM_CHUNK_ROWS", str(chunk))
        first = ugs_sampler.sample_batch(ei_t, ptr_t, m, k, mode=mode, seed=7)          # first call of a shape: two-phase, remembers the total
        key = (ei.shape[1], len(ptr) - 1, m, k, mode)
        assert ugs_sampler._stream_totals[key] >= two_phase[1].shape[1]                   # (the largest total seen for the shape)
        streamed = ugs_sampler._sample_batch_streamed(ei_t.data_ptr(), ei_t.stride(0), ei.shape[1], ptr_t, len(ptr) - 1, m, k, mode, 7)
        assert streamed is not None
        other_seed = ugs_sampler.sample_batch(ei_t, ptr_t, m, k, mode=mode, seed=8)
        want = oracle.sample_batch(ei, ptr, m, k, mode, 7, cache=cache)
        want8 = oracle.sample_batch(ei, ptr, m, k, mode, 8, cache=cache)
        for name, a, b, c, w, o8, w8 in zip(("nodes", "edge_index", "edge_ptr", "sample_ptr", "edge_src"), two_phase, first, streamed, want, other_seed, want8):
            assert c.is_contiguous() and c.dtype == torch.int64
            assert np.array_equal(a.numpy(), np.asarray(w)), (shape, mode, name, "two-phase vs oracle")
            assert np.array_equal(b.numpy(), np.asarray(w)) and np.array_equal(c.numpy(), np.asarray(w)), (shape, mode, name, "streamed vs oracle")
            assert np.array_equal(o8.numpy(), np.asarray(w8)), (shape, mode, name, "streamed, other seed vs oracle")
    # early start: the streamed calls above began their walks before the content hash was done, and kept them; a batch that agrees
    # with a remembered one in the 32 sampled words but not elsewhere starts on the wrong plan, is found out and runs again
    def stats():
        a, b = C.c_int64(), C.c_int64()
        assert lib.ugs_stream_stats(C.byref(a), C.byref(b)) == 0
        return a.value, b.value
    kept0, wrong0 = stats()
    assert kept0 >= 4
    E = ei.shape[1]
    sampled = {i * (E - 1) // 13 for i in range(14)}
    c = next(j for j in range(E // 2, E) if j not in sampled and j - 1 not in sampled and tuple(ei[:, j]) != tuple(ei[:, j - 1]))
    ei2 = ei.copy()
    ei2[:, c] = ei2[:, c - 1]                                                            # one column replaced by a copy of its neighbour
    ei2_t = torch.from_numpy(ei2)
    monkeypatch.setenv("UGS_STREAM_CHUNK_ROWS", "211")
    got = ugs_sampler._sample_batch_streamed(ei2_t.data_ptr(), ei2_t.stride(0), E, ptr_t, len(ptr) - 1, m, k, "sample", 7)
    # (a graph of more than 1000 columns is keyed by SAMPLED columns in the reference's LRU, include/cache.hpp:81-109: there the changed
    # batch can be the old graph as far as the cache is concerned, the lookup names the same plan, and the oracle agrees)
    kept1, wrong1 = stats()
    assert got is not None and kept1 + wrong1 == kept0 + wrong0 + 1 and (shape != "small_graphs" or wrong1 == wrong0 + 1)
    for a, w in zip(got, oracle.sample_batch(ei2, ptr, m, k, "sample", 7, cache=cache)):
        assert np.array_equal(a.numpy(), np.asarray(w))
    got = ugs_sampler._sample_batch_streamed(ei2_t.data_ptr(), ei2_t.stride(0), E, ptr_t, len(ptr) - 1, m, k, "sample", 7)
    assert got is not None and stats() == (kept1 + 1, wrong1)                            # now it is the remembered one (most recent first)
    for a, w in zip(got, oracle.sample_batch(ei2, ptr, m, k, "sample", 7, cache=cache)):
        assert np.array_equal(a.numpy(), np.asarray(w))
    monkeypatch.setenv("UGS_NO_SPECULATION", "1")
    got = ugs_sampler._sample_batch_streamed(ei_t.data_ptr(), ei_t.stride(0), E, ptr_t, len(ptr) - 1, m, k, "sample", 7)
    assert stats() == (kept1 + 1, wrong1)
    for a, b in zip(got, two_phase):
        assert np.array_equal(a.numpy(), b.numpy())
    monkeypatch.delenv("UGS_NO_SPECULATION")
    # the two-phase call starts early too for large batches (threshold lowered here): a kept start, a thrown-away one, host and device outputs
    monkeypatch.setenv("UGS_NO_STREAMED_CALL", "1")
    monkeypatch.setenv("UGS_SPEC_MIN_COLS", "1")
    ei3 = ei.copy()
    ei3[:, c] = ei3[:, c + 1] if c + 1 not in sampled else ei3[:, c - 2]
    k_a, w_a = stats()
    for dev in (None, "cuda:0"):
        for arr, seed in ((ei, 11), (ei3, 11), (ei3, 12)):
            got = ugs_sampler.sample_batch(torch.from_numpy(arr), ptr_t, m, k, mode="graph", seed=seed, device=dev)
            for a, w in zip(got, oracle.sample_batch(arr, ptr, m, k, "graph", seed, cache=cache)):
                assert np.array_equal(a.cpu().numpy(), np.asarray(w))
    k_b, w_b = stats()
    assert k_b + w_b == k_a + w_a + 6 and k_b >= k_a + 3 and (shape != "small_graphs" or w_b >= w_a + 2)
    monkeypatch.delenv("UGS_SPEC_MIN_COLS")
    monkeypatch.delenv("UGS_NO_STREAMED_CALL")
    # the estimate exactly at the total works; one below it is refused with UGS_E_CAPACITY, and the shim then takes the two-phase path
    tot = two_phase[1].shape[1]
    assert tot > 0
    for cap, expect in ((tot, 0), (tot - 1, UGS_E_CAPACITY)):
        bufs = [torch.empty(n, dtype=torch.int64).pin_memory() for n in (B * k, 2 * max(cap, 1), B + 1, len(ptr), max(cap, 1))]
        t = C.c_int64()
        rc = lib.ugs_sample_batch_stream(ei_t.data_ptr(), ei_t.stride(0), ei.shape[1], ptr_t.data_ptr(), len(ptr) - 1, m, k, 0, 7, cap,
                                         *[b.data_ptr() for b in bufs], C.byref(t))
        assert rc == expect, (rc, expect)
        if rc == 0:
            assert t.value == tot and np.array_equal(bufs[1][:2 * tot].view(2, tot).numpy(), two_phase[1].numpy())
    key = (ei.shape[1], len(ptr) - 1, m, k, "sample")
    ugs_sampler._stream_totals[key] = tot // 2                                           # an estimate far too small
    again = ugs_sampler.sample_batch(ei_t, ptr_t, m, k, mode="sample", seed=7)
    assert ugs_sampler._stream_totals[key] == tot                                        # the two-phase path raised it
    for a, b in zip(again, two_phase):
        assert np.array_equal(a.numpy(), b.numpy())
    cache.close()
    ugs_sampler._stream_totals.clear()


def test_streamed_handle_call_equals_the_two_phase_call_and_the_oracle(monkeypatch):
    """ugs_sample_stream (sample() of the handle API in chunks with the copy-out beside the walks): every edge mode incl. a base offset,
    chunk sizes that leave a last chunk of one row, against the two-phase call and the CPU oracle; an estimate too small falls back."""
    import torch
    import oracle
    import ugs_sampler
    rng = np.random.default_rng(11)
    n, m, k = 2500, 1801, 7
    ei = np.ascontiguousarray(_er(rng, n, 24))
    ei_t = torch.from_numpy(ei)
    h = ugs_sampler.create_preproc(ei_t, n, k)
    P = oracle.Preproc(ei, n, k)
    monkeypatch.setattr(ugs_sampler, "_STREAM_MIN_ROWS", 1)
    ugs_sampler._stream_totals.clear()
    try:
        for edge_mode, off, chunk in (("local", 0, 100), ("flat", 0, 1800), ("global", 12345, 601), ("local", 0, 5000)):
            monkeypatch.setenv("UGS_STREAM_CHUNK_ROWS", str(chunk))
            monkeypatch.setenv("UGS_NO_STREAMED_CALL", "1")
            two_phase = ugs_sampler.sample(h, m, k, edge_mode, off, 3)
            monkeypatch.delenv("UGS_NO_STREAMED_CALL")
            first = ugs_sampler.sample(h, m, k, edge_mode, off, 3)                     # learns the total
            key = ("handle", int(h), m, k, edge_mode)
            assert ugs_sampler._stream_totals[key] >= two_phase[1].shape[1]
            streamed = ugs_sampler.sample(h, m, k, edge_mode, off, 3)
            other = ugs_sampler.sample(h, m, k, edge_mode, off, 4)
            want, want4 = P.sample(m, k, edge_mode, off, 3), P.sample(m, k, edge_mode, off, 4)
            for name, a, b, c, w, o, w4 in zip(("nodes", "edge_index", "edge_ptr", "edge_src"), two_phase, first, streamed, want, other, want4):
                assert c.is_contiguous() and c.dtype == torch.int64 and c.is_pinned()
                assert np.array_equal(a.numpy(), np.asarray(w)) and np.array_equal(b.numpy(), np.asarray(w)), (edge_mode, name)
                assert np.array_equal(c.numpy(), np.asarray(w)) and np.array_equal(o.numpy(), np.asarray(w4)), (edge_mode, name, "streamed")
        ugs_sampler._stream_totals[("handle", int(h), m, k, "local")] = 10              # far too small: the two-phase path serves the call
        again = ugs_sampler.sample(h, m, k, "local", 0, 3)
        for a, w in zip(again, P.sample(m, k, "local", 0, 3)):
            assert np.array_equal(a.numpy(), np.asarray(w))
        assert ugs_sampler._stream_totals[("handle", int(h), m, k, "local")] == again[1].shape[1]
    finally:
        P.close()
        ugs_sampler.destroy_preproc(h)
        ugs_sampler._stream_totals.clear()


def test_streamed_call_edge_cases_vs_oracle():
    """ugs_sample_batch_stream on the inputs the reference's tests treat specially: no graphs, m = 0, k = 1 (no edges: capacity 0 and
    null edge buffers are fine), graphs smaller than k (rows of -1), a batch without columns -- against the oracle.  The graph cache is
    emptied before every call: its key ignores k as the reference's does, and the oracle call starts from a fresh one."""
    import ctypes as C
    import torch
    import oracle
    import ugs_sampler
    from ugs_sampler._lib import lib, UGS_E_CAPACITY

    def stream(ei, ptr, m, k, cap, mode=0, seed=5):
        ei = np.ascontiguousarray(ei.reshape(2, -1))
        G = len(ptr) - 1
        B = G * m
        nodes, eptr, sptr = torch.full((max(B * k, 1),), 7, dtype=torch.int64).pin_memory(), torch.full((B + 1,), 7, dtype=torch.int64).pin_memory(), \
            torch.full((G + 1,), 7, dtype=torch.int64).pin_memory()
        eidx, esrc = torch.full((max(2 * cap, 1),), 7, dtype=torch.int64).pin_memory(), torch.full((max(cap, 1),), 7, dtype=torch.int64).pin_memory()
        t = C.c_int64(-1)
        ei_t, ptr_t = torch.from_numpy(ei), torch.from_numpy(np.asarray(ptr, dtype=np.int64))
        rc = lib.ugs_sample_batch_stream(ei_t.data_ptr() if ei.size else None, ei.shape[1], ei.shape[1], ptr_t.data_ptr(), G, m, k, mode, seed, cap,
                                         nodes.data_ptr(), eidx.data_ptr() if cap else None, eptr.data_ptr(), sptr.data_ptr(),
                                         esrc.data_ptr() if cap else None, C.byref(t))
        if rc != 0:
            return rc, t.value                                            # (a capacity error reports the entries it had reached)
        return rc, t.value, nodes[:B * k].view(B, k).numpy(), eidx[:2 * max(t.value, 0)].view(2, -1).numpy(), eptr.numpy(), sptr.numpy(), esrc[:max(t.value, 0)].numpy()

    def check(ei, ptr, m, k, cap):
        ugs_sampler.clear_cache()
        rc, tot, nodes, eidx, eptr, sptr, esrc = stream(ei, ptr, m, k, cap)
        assert rc == 0
        w = [np.asarray(x) for x in oracle.sample_batch(np.ascontiguousarray(ei.reshape(2, -1)), np.asarray(ptr, dtype=np.int64), m, k, "sample", 5)]
        assert tot == w[1].shape[1]
        for got, want in zip((nodes, eidx, eptr, sptr, esrc), w):
            assert np.array_equal(got, want.reshape(got.shape))

    tri = np.array([[0, 1, 2, 3, 4], [1, 2, 0, 4, 3]], dtype=np.int64)                  # a triangle and an edge
    none = np.zeros((2, 0), dtype=np.int64)
    check(none, [0], 4, 3, 0)                                                           # no graphs
    check(tri, [0, 3, 5], 0, 3, 0)                                                      # m = 0
    check(tri, [0, 3, 5], 9, 1, 0)                                                      # k = 1: nodes only
    check(tri, [0, 3, 5], 9, 3, 200)                                                    # the second graph is smaller than k: rows of -1
    check(none, [0, 4], 5, 2, 0)                                                        # a graph without columns
    check(tri, [0, 3, 5], 9, 2, 200)
    ugs_sampler.clear_cache()
    rc, reached = stream(tri, [0, 3, 5], 9, 2, 0)                                       # edges but no room for them
    assert rc == UGS_E_CAPACITY and reached > 0
    assert stream(tri, [0, 3, 5], 9, 2, 200, mode=5)[0] != 0 and b"mode must be one of" in lib.ugs_last_error()


def test_streamed_handle_call_edge_cases_vs_oracle():
    """ugs_sample_stream on the handle API's special inputs: m = 0, k = 1 (capacity 0, null edge buffers), every edge mode on a small
    graph, an unknown handle, a bad edge mode, and a capacity too small (the call reports the entries it had reached)."""
    import ctypes as C
    import torch
    import oracle
    import ugs_sampler
    from ugs_sampler._lib import lib, UGS_E_CAPACITY

    ei = np.array([[0, 1, 2, 3, 4, 0, 2], [1, 2, 0, 4, 5, 3, 5]], dtype=np.int64)        # a triangle with a tail, six vertices
    n = 6

    def stream(h, m, k, mode, off, cap, seed=9):
        nodes, eptr = torch.full((max(m * k, 1),), 7, dtype=torch.int64).pin_memory(), torch.full((m + 1,), 7, dtype=torch.int64).pin_memory()
        eidx, esrc = torch.full((max(2 * cap, 1),), 7, dtype=torch.int64).pin_memory(), torch.full((max(cap, 1),), 7, dtype=torch.int64).pin_memory()
        t = C.c_int64(-1)
        rc = lib.ugs_sample_stream(h, m, k, mode, off, seed, cap, nodes.data_ptr(), eidx.data_ptr() if cap else None, eptr.data_ptr(),
                                   esrc.data_ptr() if cap else None, C.byref(t))
        if rc != 0:
            return rc, t.value
        return rc, t.value, nodes[:m * k].view(m, k).numpy(), eidx[:2 * t.value].view(2, -1).numpy(), eptr.numpy(), esrc[:t.value].numpy()

    for k in (1, 2, 3, 6):
        h = ugs_sampler.create_preproc(torch.from_numpy(ei), n, k)
        P = oracle.Preproc(ei, n, k)
        try:
            for name, mode, off in (("local", 0, 0), ("flat", 1, 0), ("global", 2, 777)):
                for m in (0, 1, 23):
                    cap = 0 if k == 1 or m == 0 else 2 * m * k * k
                    rc, tot, *got = stream(h, m, k, mode, off, cap)
                    assert rc == 0, (k, name, m, lib.ugs_last_error())
                    want = [np.asarray(x) for x in P.sample(m, k, name, off, 9)]
                    assert tot == want[1].shape[1]
                    for g, w in zip(got, want):
                        assert np.array_equal(g, w.reshape(g.shape)), (k, name, m)
            if k == 3:
                rc, reached = stream(h, 23, k, 0, 0, 1)
                assert rc == UGS_E_CAPACITY and reached > 1
                assert stream(h, 23, k, 4, 0, 500)[0] != 0 and b"edge_mode must be one of" in lib.ugs_last_error()
        finally:
            P.close()
            ugs_sampler.destroy_preproc(h)
    assert stream(987654321, 4, 2, 0, 0, 100)[0] != 0 and b"Invalid preproc handle" in lib.ugs_last_error()
