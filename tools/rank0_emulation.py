"""Rank 0 of an 8-rank strong-scaled C5 job emulated on ONE GPU: its shard's walk/scan/fill on the main stream, and beside it the
collation of the WHOLE 1M-row batch (pack of the own shard, a device copy standing in for the 7 received messages, unpack of 8
messages) on a side stream, double-buffered like bench.py.  Reports steady-state ms per step for walk shares and shard sizes."""
import sys, time
sys.path.insert(0, "ss-gnn_amd")
import torch, numpy as np, ugs_sampler, ugs_workloads as wl
from ugs_sampler import distributed as ud
ei, ptr, m, k = wl.workload("c5_er_1m")
dev = torch.device("cuda:0")
plan = ugs_sampler.Plan.from_batch(torch.from_numpy(ei), torch.from_numpy(ptr), k)
total = m
def run(share, rows, steps=24, collate=True):
    plan.set_walk_share(share)
    nodes0, eptr0, tot = plan.walk(m, "sample", 1, 0, rows, sync=True)
    cap = int(tot * 1.15) + 1024
    col = ud.Collator(total, k, "sample", int(ptr[-1]), max(int(ptr[-1]), m * k), ei.shape[1], int(cap * (total / 8) / rows) + 1024, dev, world=8, rank=0)
    rows_c = col.rows                                               # the collator's own (even) shard: the message format is fixed
    bufs = []
    for b in range(2):
        n = torch.empty((rows_c, k), dtype=torch.int64, device=dev); p = torch.empty((rows_c + 1,), dtype=torch.int64, device=dev)
        e = torch.empty((2, col.edge_cap), dtype=torch.int64, device=dev); s = torch.empty((col.edge_cap,), dtype=torch.int64, device=dev)
        bufs.append((n, p, e, s))
    side = torch.cuda.Stream()
    main = torch.cuda.current_stream()
    ev_fill = [torch.cuda.Event(), torch.cuda.Event()]; ev_col = [torch.cuda.Event(), torch.cuda.Event()]
    def step(i):
        b = i & 1
        n, p, e, s = bufs[b]
        main.wait_event(ev_col[b])                                  # the collation that read these buffers two steps ago
        # the shard that is WALKED has `rows` rows (uneven split); what is packed is the collator's fixed shard shape
        plan.walk(m, "sample", 100 + i, 0, rows, out=(n[:rows] if rows <= rows_c else None, p[:rows + 1] if rows <= rows_c else None), sync=False) if rows <= rows_c else plan.walk(m, "sample", 100 + i, 0, rows_c, out=(n, p), sync=False)
        plan.fill(m, n[:min(rows, rows_c)], p[:min(rows, rows_c) + 1], None, "sample", 0, out=(e, s))
        ev_fill[b].record(main)
        if collate:
            with torch.cuda.stream(side):
                side.wait_event(ev_fill[b])
                col.pack((n, e, p, s))
                col.inbox.copy_(col.msg.unsqueeze(0).expand(8, -1))   # stands in for the 7 messages arriving over xGMI
                col.unpack()
                ev_col[b].record(side)
    for i in range(4): step(i)
    torch.cuda.synchronize(); t = time.perf_counter()
    for i in range(4, 4 + steps): step(i)
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / steps * 1e3
for share in (100, 80):
    for rows in (125000, 100000):
        print(f"share {share:3d}  rows {rows}:  walk-only {run(share, rows, collate=False):.3f} ms/step   with the batch's collation {run(share, rows):.3f} ms/step", flush=True)
