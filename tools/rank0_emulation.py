"""The two kinds of rank of an 8-rank strong-scaled C5 job (BASELINE config 5: 1M rows split over 8 GPUs), each emulated on ONE GPU:

  destination (rank 0)  its shard's walk / scan / fill, the unpack of the PREVIOUS 1M-row batch (8 messages) and the pack of its own
                        shard, all on the main stream like bench.py's Job; on a side stream a device copy standing in for the 7
                        messages arriving over xGMI while the next batch is sampled;
  other ranks           their shard's walk / scan / fill and the pack of their message (the send itself is a DMA).

The job's step time is the slower of the two.  Run for the OLD split (equal shards, every rank's walk on 80 % of each CU: round 2)
and for destination-aware splits (rank 0 fewer rows, the others the whole CU): writes gpurun_out/r03_rank0_emulation.json, which is
kept under profiles/.  usage: python tools/rank0_emulation.py [--world 8]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ss-gnn_amd"))
import torch                      # noqa: E402
import ugs_sampler                # noqa: E402
import ugs_workloads as wl        # noqa: E402
from ugs_sampler import distributed as ud   # noqa: E402

WORLD = int(sys.argv[sys.argv.index("--world") + 1]) if "--world" in sys.argv else 8
ei, ptr, m, k = wl.workload("c5_er_1m")
dev = torch.device("cuda:0")
plan = ugs_sampler.Plan.from_batch(torch.from_numpy(ei), torch.from_numpy(ptr), k)
total = m
node_bound = int(ptr[-1])


twin = plan.twin()


def run(share, rows0, rank, steps=24, collate=True, nstreams=2):
    """steady-state ms per step of `rank` (0 = destination) when rank 0 samples rows0 rows and the others share the rest equally"""
    w0 = rows0 * (WORLD - 1) / (total - rows0)
    row_off = ud.shard_offsets(total, WORLD, [w0] + [1.0] * (WORLD - 1))
    begin, rows = row_off[rank], row_off[rank + 1] - row_off[rank]
    plan.set_walk_share(share)
    twin.set_walk_share(share)
    _, _, tot = plan.walk(m, "sample", 1, begin, rows, sync=True)
    rows_cap = max(row_off[r + 1] - row_off[r] for r in range(WORLD))
    cap = int(tot * 1.1 * rows_cap / rows) + 4096
    col = ud.Collator(total, k, "sample", node_bound, max(node_bound, m * k), ei.shape[1], cap, dev, world=WORLD, rank=rank, dst=0, row_off=row_off)
    sets = [(torch.empty((rows, k), dtype=torch.int64, device=dev), torch.empty((rows + 1,), dtype=torch.int64, device=dev),
             torch.empty((2, cap), dtype=torch.int64, device=dev), torch.empty((cap,), dtype=torch.int64, device=dev)) for _ in range(nstreams)]
    side, main = torch.cuda.Stream(), torch.cuda.current_stream()
    streams = [main] + [torch.cuda.Stream() for _ in range(nstreams - 1)]
    plans = [plan, twin][:nstreams]
    ev_packed, ev_exchanged = torch.cuda.Event(), torch.cuda.Event()
    state = {"in_flight": False}
    if rank == 0:                                    # realistic inbox: WORLD valid messages (the other ranks' shards are about this large)
        n, p, e, s = sets[0]
        plan.walk(m, "sample", 7, begin, rows, out=(n, p), sync=False)
        plan.fill(m, n, p, None, "sample", begin, out=(e, s))
        col.pack((n, e, p, s))
        col.inbox.copy_(col.msg.unsqueeze(0).expand(WORLD, -1))
        torch.cuda.synchronize()

    def step(i):                                     # bench.py's Job: consecutive steps alternate between the streams; only the exchange on the side stream
        b = i % nstreams
        n, p, e, s = sets[b]
        st = streams[b]
        with torch.cuda.stream(st):
            plans[b].walk(m, "sample", 100 + i, begin, rows, out=(n, p), sync=False)
            plans[b].fill(m, n, p, None, "sample", begin, out=(e, s))
            if not collate:
                return
            if state["in_flight"]:
                st.wait_event(ev_exchanged)
                if rank == 0:
                    col.unpack()                     # the previous batch: its messages arrived while this one was sampled
            col.pack((n, e, p, s))
            ev_packed.record(st)
        with torch.cuda.stream(side):
            side.wait_event(ev_packed)
            if rank == 0:
                col.inbox[1:].copy_(col.msg.unsqueeze(0).expand(WORLD - 1, -1))   # stands in for the WORLD-1 messages arriving over xGMI
            ev_exchanged.record(side)
        state["in_flight"] = True

    for i in range(4):
        step(i)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for i in range(4, 4 + steps):
        step(i)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t) / steps * 1e3
    assert not col.overflowed()
    return round(ms, 4), rows


out = {"workload": "c5_er_1m", "world": WORLD, "global_rows": total,
       "stream_plan": "two steps in flight on two streams (plan + twin), collation work on the step's stream, exchange on a side stream (round 3)", "cases": []}
equal = total // WORLD
# old split (round 2): equal shards, every rank's walk at 80 %
d_ms, d_rows = run(80, equal, 0)
o_ms, o_rows = run(80, equal, 1)
out["old_equal_split_share80_everywhere"] = {"rank0_rows": d_rows, "rank0_ms": d_ms, "other_rows": o_rows, "other_ms": o_ms, "step_ms": max(d_ms, o_ms),
                                             "k_subgraphs_per_s": round(total / max(d_ms, o_ms) * 1e3, 1)}
print(json.dumps(out["old_equal_split_share80_everywhere"]), flush=True)
for share in (80, 100):
    for frac in (1.0, 0.9, 0.8, 0.75, 0.7, 0.65):
        rows0 = int(equal * frac)
        d_ms, d_rows = run(share, rows0, 0)
        d_alone, _ = run(share, rows0, 0, collate=False)
        o_ms, o_rows = run(100, rows0, 1)
        case = {"rank0_walk_share": share, "rank0_rows": d_rows, "rank0_ms": d_ms, "rank0_ms_without_collation": d_alone, "other_rows": o_rows,
                "other_ms_share100": o_ms, "step_ms": max(d_ms, o_ms), "k_subgraphs_per_s": round(total / max(d_ms, o_ms) * 1e3, 1)}
        out["cases"].append(case)
        print(json.dumps(case), flush=True)
best = min(out["cases"], key=lambda c: c["step_ms"])
out["best"] = best
# bench.py's --dst-rows auto, replayed: every round measures each kind of rank on its own, next weights = rows per millisecond
for share in (80, 100):
    rows0, rounds = equal, []
    for _ in range(3):
        d_ms, d_rows = run(share, rows0, 0, steps=48)
        o_ms, o_rows = run(100, rows0, 1, steps=48)
        rounds.append({"rank0_rows": d_rows, "rank0_ms": d_ms, "other_rows": o_rows, "other_ms": o_ms})
        w0, w1 = d_rows / d_ms, o_rows / o_ms
        rows0 = int(total * w0 / (w0 + (WORLD - 1) * w1))
    d_ms, d_rows = run(share, rows0, 0, steps=96)
    o_ms, o_rows = run(100, rows0, 1, steps=96)
    d1, _ = run(share, rows0, 0, nstreams=1)
    o1, _ = run(100, rows0, 1, nstreams=1)
    out[f"auto_split_share{share}"] = {"calibration_rounds": rounds, "rank0_rows": d_rows, "rank0_ms": d_ms, "other_rows": o_rows, "other_ms": o_ms,
                                       "step_ms": max(d_ms, o_ms), "k_subgraphs_per_s": round(total / max(d_ms, o_ms) * 1e3, 1),
                                       "same_split_one_step_at_a_time": {"rank0_ms": d1, "other_ms": o1, "step_ms": max(d1, o1)}}
    print("auto", share, json.dumps(out[f"auto_split_share{share}"]), flush=True)
out["note"] = ("one GPU plays one rank at a time; the 7 incoming messages are a device copy; whether RCCL's receive kernels find wave slots beside "
               "the walk is not covered (needs a multi-GPU node); bench.py runs these three calibration rounds")
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
with open(os.path.join(ROOT, "gpurun_out", "r03_rank0_emulation.json"), "w") as f:
    json.dump(out, f, indent=1)
print("best:", json.dumps(best))
