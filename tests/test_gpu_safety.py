"""GPU: memory- and stream-safety of the boundary (regressions for review findings, no reference counterpart).

  * fill kernels never write past the capacity the caller states (`ld`), whatever the batch's multiplicity;
  * jobs that return device tensors run on torch's current stream (ugs_set_stream), so they are ordered with the consumer;
  * one plan used from two streams: the second call waits for the first one's kernels (the plan's scratch is shared).
Everything is checked against the CPU oracle (bit-exact, integer outputs)."""
import numpy as np
import pytest

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(180)]


def _clique_batch(n, G, copies=1):
    """G cliques of n vertices, both directions of every edge stored (PyG style), every column `copies` times."""
    cols, ptr = [], [0]
    for g in range(G):
        off = g * n
        e = [(off + u, off + v) for u in range(n) for v in range(n) if u != v]
        cols += e * copies
        ptr.append(off + n)
    return np.array(cols, dtype=np.int64).T.reshape(2, -1).copy(), np.array(ptr, dtype=np.int64)


def test_captured_step_never_writes_past_its_capacity():
    import torch
    import oracle
    import ugs_sampler
    k, m = 4, 16
    # (a) both directions stored: every ordered pair of a row shows up twice -> 2*k*(k-1) entries per row, the default capacity
    ei, ptr = _clique_batch(6, 4)
    plan = ugs_sampler.Plan.from_batch(torch.from_numpy(ei), torch.from_numpy(ptr), k)
    rows = 4 * m
    step = plan.graph_step(m, "sample")
    assert step.capacity == 2 * rows * k * (k - 1)
    step.launch(7)
    got = [t.cpu().numpy() for t in step.result()]
    want = oracle.sample_batch(ei, ptr, m, k, "sample", 7)
    assert int(want[2][-1]) == step.capacity                                  # the bound is tight for cliques
    for a, b in zip(got, (want[0], want[1], want[2], want[4])):
        assert np.array_equal(a, b)
    step.close()
    # (b) repeated columns exceed any k-bound: a too small capacity must leave everything behind it untouched and be reported
    ei3, ptr3 = _clique_batch(6, 4, copies=3)
    plan3 = ugs_sampler.Plan.from_batch(torch.from_numpy(ei3), torch.from_numpy(ptr3), k)
    want3 = oracle.sample_batch(ei3, ptr3, m, k, "sample", 7)
    total = int(want3[2][-1])
    cap = total // 2
    dev = torch.device("cuda", torch.cuda.current_device())
    CAN = -7777
    big_idx = torch.full((2, cap + 4096), CAN, dtype=torch.int64, device=dev)
    big_src = torch.full((cap + 4096,), CAN, dtype=torch.int64, device=dev)
    nodes, eptr, tot = plan3.walk(m, "sample", 7)
    assert tot == total
    # edge_index [2, cap] as a strided view of a wider canary buffer; edge_src [cap] likewise
    from ugs_sampler._lib import check, lib
    check(lib.ugs_plan_fill(plan3._h, m, k, 0, 0, 0, rows, torch.cuda.current_stream().cuda_stream, nodes.data_ptr(), eptr.data_ptr(),
                            big_idx.data_ptr(), cap, big_src.data_ptr()))
    torch.cuda.synchronize()
    flat = big_idx.reshape(-1).cpu().numpy()
    assert np.array_equal(flat[:cap], want3[1][0, :cap]) and np.array_equal(flat[cap:2 * cap], want3[1][1, :cap])
    assert (flat[2 * cap:] == CAN).all(), "fill wrote past the stated capacity of edge_index"
    src = big_src.cpu().numpy()
    assert np.array_equal(src[:cap], want3[4][:cap]) and (src[cap:] == CAN).all(), "fill wrote past the stated capacity of edge_src"
    step3 = plan3.graph_step(m, "sample", edge_capacity=cap)
    step3.launch(7)
    with pytest.raises(RuntimeError, match="edge capacity"):
        step3.result()
    step3.close()


def test_device_outputs_are_ordered_with_torchs_current_stream():
    import torch
    import oracle
    import ugs_sampler
    import ugs_workloads as wl
    ei, ptr = wl.tu_batch(39, 73, 8)
    ei_t, ptr_t = torch.from_numpy(ei), torch.from_numpy(ptr)
    want = oracle.sample_batch(ei, ptr, 64, 6, "sample", 42)
    dev = torch.device("cuda", torch.cuda.current_device())
    side = torch.cuda.Stream(device=dev)
    x = torch.ones((4096, 4096), device=dev)
    for _ in range(3):
        with torch.cuda.stream(side):
            # a long-running reader of freshly freed blocks is queued on `side`; the job must queue behind it on the same stream
            junk = [torch.empty((want[0].shape[0], 6), dtype=torch.int64, device=dev) for _ in range(4)]
            y = x @ x
            for j in junk:
                j.fill_(-1)
            del junk
            got = ugs_sampler.sample_batch(ei_t, ptr_t, 64, 6, mode="sample", seed=42, device=dev)
            chk = [g.clone() for g in got]                   # consumer on the same stream: must see the job's output
        side.synchronize()
        for a, b in zip(chk, want):
            assert np.array_equal(a.cpu().numpy(), b)
        del y
    # host outputs afterwards use the library's own stream again
    got = ugs_sampler.sample_batch(ei_t, ptr_t, 64, 6, mode="sample", seed=42)
    for a, b in zip(got, want):
        assert np.array_equal(a.numpy(), b)


def test_one_plan_on_two_streams():
    import torch
    import oracle
    import ugs_sampler
    import ugs_workloads as wl
    ei, ptr = wl.er_graph(20000, 400000, 3)
    ei_t, ptr_t = torch.from_numpy(ei), torch.from_numpy(ptr)
    k, m = 6, 30000
    plan = ugs_sampler.Plan.from_batch(ei_t, ptr_t, k)
    P = oracle.Preproc(ei, 20000, k)
    dev = torch.device("cuda", torch.cuda.current_device())
    s1, s2 = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
    outs = {}
    for rep in range(3):
        for seed, st in ((11 + rep, s1), (501 + rep, s2)):       # back to back, no host synchronisation in between
            with torch.cuda.stream(st):
                nodes, eptr, _ = plan.walk(m, "sample", seed, sync=False)
                outs[seed] = (nodes, eptr)
    torch.cuda.synchronize()
    for seed, (nodes, eptr) in outs.items():
        w = P.sample(m, k, "local", 0, seed)
        assert np.array_equal(nodes.cpu().numpy(), w[0]) and np.array_equal(eptr.cpu().numpy(), w[2]), f"seed {seed}"
    # a sample_batch job (library / torch stream) right after an asynchronous plan call on another stream
    with torch.cuda.stream(s1):
        nodes, eptr, _ = plan.walk(m, "sample", 77, sync=False)
    got = ugs_sampler.sample_batch(ei_t, ptr_t, 2000, k, mode="sample", seed=5)
    torch.cuda.synchronize()
    w = P.sample(m, k, "local", 0, 77)
    assert np.array_equal(nodes.cpu().numpy(), w[0]) and np.array_equal(eptr.cpu().numpy(), w[2])
    w5 = oracle.sample_batch(ei, ptr, 2000, k, "sample", 5)
    for a, b in zip(got, w5):
        assert np.array_equal(a.numpy(), b)


@pytest.mark.parametrize("mode,world,weights", [("sample", 3, None), ("global", 2, None), ("graph", 5, None),
                                                ("sample", 3, [0.4, 1.0, 1.2]), ("global", 2, [0.0, 1.0]), ("graph", 5, [0.7, 1, 1, 1.1, 0.9])])
def test_collation_kernels_equal_the_torch_path_and_the_oracle(mode, world, weights):
    """ugs_collate_unpack (HIP) against the torch-operation path of the same Collator and against the single-process result:
    `world` ranks' messages are packed on the GPU, placed in the destination's inbox by hand (no process group needed), unpacked.
    Equal and uneven (destination-aware) row ranges; then a capacity below one rank's total: reported, nothing written out of bounds."""
    import torch
    import oracle
    import ugs_workloads as wl
    from ugs_sampler import distributed as ud
    ei, ptr = wl.tu_batch(18, 20, 6)
    m, k = 9, 4
    G = len(ptr) - 1
    dev = torch.device("cuda", torch.cuda.current_device())
    node_bound = int(ptr[-1])
    for seed in (1, 2):
        full = oracle.sample_batch(ei, ptr, m, k, mode, seed)
        nodes, eidx, eptr, _, esrc = full
        row_off = ud.shard_offsets(G * m, world, weights)
        spans = [(row_off[r], row_off[r + 1] - row_off[r]) for r in range(world)]
        cap = max(int(eptr[b + c] - eptr[b]) for b, c in spans) + 5
        mk = lambda r, d, cap_=cap: ud.Collator(G * m, k, mode, node_bound, max(node_bound, m * k), ei.shape[1], cap_, d, dst=0, world=world, rank=r,
                                                row_off=row_off)
        dst_gpu, dst_cpu = mk(0, dev), mk(0, "cpu")
        for r, (b, c) in enumerate(spans):
            e0, e1 = int(eptr[b]), int(eptr[b + c])
            junk = np.full((2, cap - (e1 - e0)), -9, dtype=np.int64)                      # capacity slack beyond the rank's total
            local = (torch.from_numpy(nodes[b:b + c].copy()), torch.from_numpy(np.concatenate([eidx[:, e0:e1], junk], axis=1)),
                     torch.from_numpy((eptr[b:b + c + 1] - e0).copy()), torch.from_numpy(np.concatenate([esrc[e0:e1], junk[0]])))
            msg = mk(r, dev).pack(tuple(t.to(dev) for t in local))
            dst_gpu.inbox[r].copy_(msg)
            dst_cpu.inbox[r].copy_(msg.cpu())
        got = [t.cpu().numpy() for t in dst_gpu.unpack()]
        ref = [t.numpy() for t in dst_cpu.unpack()]
        tot = int(eptr[-1])
        assert int(got[2][-1]) == tot == int(ref[2][-1])
        for g, r_, w in ((got[0], ref[0], nodes), (got[2], ref[2], eptr), (got[1][:, :tot], ref[1][:, :tot], eidx), (got[3][:tot], ref[3][:tot], esrc)):
            assert np.array_equal(g, w) and np.array_equal(r_, w)
        assert not dst_gpu.overflowed() and not dst_cpu.overflowed()
        # capacity 3 entries short of the fullest rank: that rank's message is truncated, the destination finds out from the headers
        short = cap - 5 - 3
        if short > 0:
            sg, sc = mk(0, dev, short), mk(0, "cpu", short)
            guard = torch.full((64,), -12345, dtype=torch.int64, device=dev)              # allocated right behind the output buffers
            for r, (b, c) in enumerate(spans):
                e0, e1 = int(eptr[b]), int(eptr[b + c])
                local = (torch.from_numpy(nodes[b:b + c].copy()), torch.from_numpy(eidx[:, e0:e1].copy()), torch.from_numpy((eptr[b:b + c + 1] - e0).copy()),
                         torch.from_numpy(esrc[e0:e1].copy()))
                packer = mk(r, dev, short)
                msg = packer.pack(tuple(t.to(dev) for t in local))
                assert packer.overflowed() == (e1 - e0 > short)
                sg.inbox[r].copy_(msg)
                sc.inbox[r].copy_(msg.cpu())
            o_g, o_c = sg.unpack(), sc.unpack()
            torch.cuda.synchronize()
            assert sg.overflowed() and sc.overflowed() and int(sg.max_total.item()) == int(sc.max_total.item()) == cap - 5
            assert np.array_equal(o_g[0].cpu().numpy(), nodes) and np.array_equal(o_c[0].numpy(), nodes)    # the rows themselves are intact
            assert bool((guard == -12345).all())
            with pytest.raises(RuntimeError):
                sg.check()


@pytest.mark.timeout(300)
def test_bench_multi_rank_control_flow_on_one_gpu():
    """`bench.py --gpus 2 --rehearse`: two ranks started by the script itself, both on this GPU, messages staged through the host
    over gloo -- every step of the N > 1 path (row shards, capacity agreement, double-buffered sampling, collation each step,
    placement check of both ranks' rows against the unsharded rows, the weak-scaling extra, orderly shutdown) runs; the numbers
    mean nothing.  The driver's scaling run is the same script over RCCL with one GPU per rank."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--rehearse", "--workload", "er_200000_4000000_100000_8",
                        "--steps", "2", "--warmup", "1", "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=280)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["scaling"] == "strong" and line["config"]["global_rows"] == 100_000
    # the destination-aware split (default --dst-rows auto): three calibration rounds, the first on the equal split; the ranks' rows cover the batch
    pr, cal = line["per_rank"], line["split_calibration"]
    assert len(cal) == 4 and cal[0]["rows"] == [50_000, 50_000] and sum(cal[1]["rows"]) == 100_000 and sum(cal[2]["rows"]) == 100_000
    assert cal[3]["split"] in ("derived from the last round", "best measured round") and len(cal[3]["slowest_rank_ms"]) == 3
    assert [p_["rank"] for p_ in pr] == [0, 1] and sum(p_["rows"] for p_ in pr) == 100_000 and line["config"]["rows_per_gpu"] == pr[0]["rows"]
    assert pr[0]["walk_share"] == 80 and pr[1]["walk_share"] == 100 and all(p_["walk_ms"] > 0 for p_ in pr)
    assert line["collate_ms_per_step"] is not None and line["extras"]["weak_scaling"]["global_rows"] == 200_000


@pytest.mark.timeout(400)
@pytest.mark.parametrize("world", [2, 3])
def test_bench_sharded_qm9_batch_on_one_gpu(world):
    """BASELINE config 4 (QM9-shaped, k = 5, 32 graphs x 2048 samples = 65 536 rows, "sharded over 8 GPUs") through the multi-rank
    path of bench.py with 2 and 3 ranks on this GPU: the row ranges cut through graphs (row b = g * m + i: a rank's range starts
    and ends inside a graph), mode "sample" sends local edge ids as uint8 on the wire, the split is calibrated (and guarded), and
    rank 0 checks the placement of its own AND the other ranks' rows against the unsharded rows.  The numbers mean nothing."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", str(world), "--rehearse", "--workload", "c4_qm9_b65536",
                        "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-extras"], env=env, capture_output=True, text=True, timeout=380)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == world and line["config"]["global_rows"] == 65_536 and line["config"]["graphs"] == 32
    pr, cal = line["per_rank"], line["split_calibration"]
    assert [p_["rank"] for p_ in pr] == list(range(world)) and sum(p_["rows"] for p_ in pr) == 65_536
    if world == 3:                                                  # m = 2048 rows per graph: no split into three ends on graph boundaries
        assert any(p_["rows"] % 2048 for p_ in pr[:-1]), "the shards were meant to cut through graphs"
    assert len(cal) == 4 and all(sum(c_["rows"]) == 65_536 for c_ in cal[:3])
    assert line["collate_ms_per_step"] is not None


@pytest.mark.parametrize("kind", ["general", "device_pass", "handle", "wave_tier"])
def test_a_plan_and_its_twin_on_two_streams_give_the_oracle_rows(kind, monkeypatch):
    """Plan.twin() (ugs_plan_twin: the same device arrays, private scratch) with consecutive seeds alternating between the plan on
    one stream and the twin on another -- plans of the general path, of the device batch pass (descriptors uploaded asynchronously:
    the twin must not run ahead of them), of a handle, and of a one-walk-per-wave tier (padded rows built before the twin copies
    the arrays); then the owner is released and the twin keeps sampling.  Every row against the oracle."""
    import torch
    import oracle
    import ugs_sampler
    import ugs_workloads as wl
    torch.cuda.set_device(0)
    ugs_sampler.clear_cache()
    k, m = 5, 40
    if kind == "wave_tier":
        ei, ptr = wl.er_graph(4000, 70000, seed=11)
        k, m = 6, 300
    else:
        ei, ptr = wl.tu_batch(18, 19, 12, dataset_seed=7)
    monkeypatch.setenv("UGS_DEVICE_BATCH", "1" if kind == "device_pass" else "0")
    if kind == "handle":
        ei, ptr = np.ascontiguousarray(ei[:, :38]), ptr[:2]          # the first graph's columns
        h = ugs_sampler.create_preproc(torch.from_numpy(ei), 18, k)
        plan = ugs_sampler.Plan.from_handle(h, k)
        pre = oracle.Preproc(ei, 18, k)
        want = lambda seed: pre.sample(m, k, "local", 0, seed)                                  # noqa: E731
        mode = "local"
    else:
        plan = ugs_sampler.Plan.from_batch(torch.from_numpy(ei), torch.from_numpy(ptr), k)
        want = lambda seed: [np.asarray(x) for i, x in enumerate(oracle.sample_batch(ei, ptr, m, k, "sample", seed)) if i != 3]   # noqa: E731
        mode = "sample"
    twin = plan.twin()
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    outs = []
    for i in range(8):
        with torch.cuda.stream(streams[i % 2]):
            outs.append((plan if i % 2 == 0 else twin).sample_rows(m, mode, 100 + i))
    torch.cuda.synchronize()
    plan.close()                                                    # the owner goes first: the arrays live while the twin does
    for i in range(8, 12):
        with torch.cuda.stream(streams[i % 2]):
            outs.append(twin.sample_rows(m, mode, 100 + i))
    torch.cuda.synchronize()
    for i, (nodes, eidx, eptr, esrc) in enumerate(outs):
        w = want(100 + i)
        assert np.array_equal(nodes.cpu().numpy(), np.asarray(w[0])) and np.array_equal(eidx.cpu().numpy(), np.asarray(w[1])), (kind, i)
        assert np.array_equal(eptr.cpu().numpy(), np.asarray(w[2])) and np.array_equal(esrc.cpu().numpy(), np.asarray(w[3])), (kind, i)
    twin.close()
    if kind == "handle":
        pre.close()
        ugs_sampler.destroy_preproc(h)
