"""CPU, multi-process (gloo): the N>1 path -- row sharding + collation of ugs_sampler.distributed -- with the CPU oracle
standing in for the per-rank row sampler (the oracle is the checker here; the shipped default row sampler is the HIP plan
path).  The collated batch on every rank must equal the single-process result bit for bit."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, mode, all_ranks, q, weights=None):
    for p in (os.path.join(ROOT, "ss-gnn_amd"), os.path.join(ROOT, "oracle")):
        sys.path.insert(0, p)
    import oracle
    import ugs_workloads as wl
    from ugs_sampler import distributed as ud
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    try:
        ei, ptr = wl.tu_batch(18, 20, 5)
        ei = np.concatenate([ei, np.array([[3], [40]], dtype=np.int64)], axis=1)      # a cross-graph column
        m, k, seed = 7, 4, 42
        full = oracle.sample_batch(ei, ptr, m, k, mode, seed)

        def row_sampler(m_, mode_, seed_, begin, count):      # stand-in: rows [begin, begin+count) of the full result
            nodes, eidx, eptr, _, esrc = full
            e0, e1 = int(eptr[begin]), int(eptr[begin + count])
            pad = np.full((2, 5), -7, dtype=np.int64)          # capacity slack beyond `total`, must be ignored
            return (torch.from_numpy(nodes[begin:begin + count].copy()),
                    torch.from_numpy(np.concatenate([eidx[:, e0:e1], pad], axis=1)),
                    torch.from_numpy((eptr[begin:begin + count + 1] - e0).copy()),
                    torch.from_numpy(np.concatenate([esrc[e0:e1], pad[0]])))

        res = ud.sample_batch_sharded(torch.from_numpy(ei), torch.from_numpy(ptr), m, k, mode=mode, seed=seed,
                                      all_ranks=all_ranks, dst=0, row_sampler=row_sampler, weights=weights)
        if res is None:
            ok = (not all_ranks) and rank != 0
        else:
            ok = all(np.array_equal(a.numpy(), b) and a.dtype == torch.int64 for a, b in zip(res, full))
        # steady state: ONE Collator, several steps with different totals, no host round trip inside collate()
        row_off = ud.shard_offsets(5 * m, world, weights)      # the destination-aware split: every rank computes the same offsets
        begin, count = row_off[rank], row_off[rank + 1] - row_off[rank]
        assert (begin, count) == ud.shard_range(5 * m, rank, world, weights)
        fulls = [oracle.sample_batch(ei, ptr, m, k, mode, s) for s in (1, 2, 3)]
        cap = max(int(f[2][begin + count] - f[2][begin]) for f in fulls)
        cap_t = torch.tensor([cap])
        dist.all_reduce(cap_t, op=dist.ReduceOp.MAX)
        node_bound = int(ptr[-1])
        col = ud.Collator(5 * m, k, mode, node_bound, max(node_bound, m * k), ei.shape[1], int(cap_t.item()) + 3, "cpu", dst=0, all_ranks=all_ranks,
                          row_off=row_off)
        for f in fulls:
            full = f
            res = col.collate(row_sampler(m, mode, 0, begin, count))
            if res is None:
                ok = ok and (not all_ranks) and rank != 0
            else:
                tot = int(res[2][-1])
                ok = ok and tot == int(f[2][-1]) and np.array_equal(res[0].numpy(), f[0]) and np.array_equal(res[2].numpy(), f[2]) \
                    and np.array_equal(res[1][:, :tot].numpy(), f[1]) and np.array_equal(res[3][:tot].numpy(), f[4])
        ok = ok and not col.overflowed()
        # a capacity below some rank's total: the step is truncated, and the collator says so (lazily, no per-step host sync)
        small = ud.Collator(5 * m, k, mode, node_bound, max(node_bound, m * k), ei.shape[1], max(int(cap_t.item()) - 2, 0), "cpu", dst=0,
                            all_ranks=all_ranks, row_off=row_off)
        for f in fulls:
            full = f
            small.collate(row_sampler(m, mode, 0, begin, count))
        flag = torch.tensor([1 if small.overflowed() else 0])
        dist.all_reduce(flag, op=dist.ReduceOp.MAX)
        ok = ok and int(flag.item()) == 1 and (small.overflowed() or not small.is_dst)
        if small.is_dst:
            try:
                small.check()
                ok = False
            except RuntimeError:
                pass
        q.put((rank, bool(ok), begin, count))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,mode,all_ranks,weights", [(2, "sample", True, None), (2, "global", False, None), (3, "graph", True, None),
                                                          (2, "sample", False, [0.6, 1.0]), (3, "global", True, [0.5, 1.0, 1.3]),
                                                          (3, "sample", False, [0.0, 1.0, 1.0])])
def test_sharded_collation_equals_single_process(world, mode, all_ranks, weights):
    """equal split and destination-aware uneven splits (rank 0, which also unpacks the batch, gets fewer rows -- down to none)"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, mode, all_ranks, q, weights)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    got = sorted(q.get(timeout=5) for _ in range(world))
    assert all(ok for _, ok, _, _ in got)
    assert [b for _, _, b, _ in got] == [sum(c for _, _, _, c in got[:i]) for i in range(world)]     # contiguous cover
    assert sum(c for _, _, _, c in got) == 5 * 7


def test_shard_range_properties():
    for p in (os.path.join(ROOT, "ss-gnn_amd"),):
        sys.path.insert(0, p)
    from ugs_sampler.distributed import shard_range
    for total in (0, 1, 7, 8, 1000003):
        for world in (1, 2, 3, 8):
            spans = [shard_range(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and sum(c for _, c in spans) == total
            assert all(spans[i][0] + spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            assert max(c for _, c in spans) - min(c for _, c in spans) <= 1


def test_weighted_shard_offsets():
    sys.path.insert(0, os.path.join(ROOT, "ss-gnn_amd"))
    from ugs_sampler.distributed import shard_offsets, shard_range

    for total in (0, 1, 35, 1_000_000, 1_000_003):
        for w in ([1, 1], [0.8, 1, 1, 1, 1, 1, 1, 1], [0, 1, 1], [3.5, 0.25, 1e-9], [1e6, 1, 1, 7]):
            off = shard_offsets(total, len(w), w)
            assert off[0] == 0 and off[-1] == total and all(a <= b for a, b in zip(off, off[1:]))
            sizes = [b - a for a, b in zip(off, off[1:])]
            exact = [total * x / sum(w) for x in w]
            assert all(abs(s_ - e) < 1.0 for s_, e in zip(sizes, exact))          # each share within one row of proportional
            assert all(shard_range(total, r, len(w), w) == (off[r], sizes[r]) for r in range(len(w)))
            if any(x == 0 for x in w):
                assert all(s_ == 0 for s_, x in zip(sizes, w) if x == 0)
    assert shard_offsets(10, 2, [1, 1]) == shard_offsets(10, 2) == [0, 5, 10]
    for bad in ([1, -1], [0, 0], [1], [float("nan"), 1], [float("inf"), 1]):
        with pytest.raises(ValueError):
            shard_offsets(10, 2, bad)
