#!/usr/bin/env python3
"""bench.py -- k-subgraphs sampled per second on MI355X (BASELINE.json metric), one JSON line on rank 0.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--scaling strong|weak]
                    [--workload c5_er_1m|c2_mutag_b1024|c3_proteins_b8192|c4_qm9_b65536|er_<n>_<cols>_<m>_<k>]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

`--gpus N` with N > 1 and no WORLD_SIZE in the environment starts the N ranks itself (fresh child processes, before anything
touches a GPU) and relays rank 0's line.

A *step* is one pass of the hot path over one batch: walk kernel(s) + scan + fill kernel, producing the reference's
tensors (nodes, edge_index, edge_ptr, edge_src) in HBM from a plan (preprocessed graph batch) that is already resident
in HBM -- the state the reference is in with a warm preprocessing LRU.  Default workload: BASELINE.json configs[4], the
Erdos-Renyi graph |V|=1M, 20M columns, k=8, batch = 1M samples.
  strong scaling (default): the N ranks split THE batch -- rank r samples rows [r*B/N, (r+1)*B/N) (legal
      because row i depends only on (seed, i), reference src/sampler.cpp:158-161) -- and the batch is collated on rank 0;
  weak scaling: every rank adds B rows of the same job (N*B rows per step), collated on rank 0.
For N > 1 every step collates the batch on rank 0 over RCCL (ugs_sampler.distributed.Collator: no host round trip): every rank
packs its rows behind its fill, the gather runs on a side stream while the next batch is sampled, and rank 0 unpacks batch s
behind the sampling of batch s+1 (class Job); all K collations are inside the timed region.  Rank 0 therefore does more per
step than the others and is given fewer rows (--dst-rows auto: measured in the warm-up).  The seed changes every step
(42 + step) so no step can reuse a previous step's output.
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "ss-gnn_amd"))

HBM_PEAK_GBPS = 8000.0   # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md, chip table)
EDGE_MODE = {"sample": "local", "graph": "flat", "global": "global"}


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="c5_er_1m")
    ap.add_argument("--mode", default="sample")
    ap.add_argument("--scaling", choices=("strong", "weak"), default="strong")
    ap.add_argument("--cpu-sample", type=int, default=-1, help="rows timed on the 1-core CPU baseline (default: sized per workload)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true")
    ap.add_argument("--no-cpu-reference", action="store_true", help="skip timing the reference C++ itself (oracle/_ref) on the headline workload; its preprocessing takes about half a minute on C5")
    ap.add_argument("--walk-share", type=int, default=80, help="DESTINATION rank: percent of each CU its walk kernels occupy, so that the unpack of the "
                    "previous batch (and RCCL's receive kernels) find room beside them (N > 1)")
    ap.add_argument("--walk-share-others", type=int, default=100, help="the other ranks only pack and send: their walk kernels keep the whole CU")
    ap.add_argument("--dst-rows", default="auto", help="N > 1: the destination's share of the rows.  'auto' (default): measured -- two short local "
                    "calibration rounds in the warm-up (three; each rank's own work per step, no exchange), one all-gather each, then rows in proportion to "
                    "rows per millisecond; 'equal': the equal split; a number w: weight of the destination against 1.0 for every other rank")
    ap.add_argument("--two-steps", action="store_true", help="N = 1 measurement aid: two steps in flight on two streams (plan + twin), as the N > 1 job "
                    "does -- the next step's walk fills this one's tail; per-kernel event times then overlap and are not a roofline figure")
    ap.add_argument("--one-stream", action="store_true", help="N > 1: one step at a time (no second stream / buffer set / plan scratch)")
    ap.add_argument("--force-collate", action="store_true", help="run the multi-GPU collation path even with one rank (rehearsal)")
    ap.add_argument("--rehearse", action="store_true", help="N ranks on ONE GPU over gloo (every rank uses cuda:0): exercises the multi-rank control flow "
                    "of this script where only one GPU is at hand; the numbers mean nothing")
    ap.add_argument("--cpu-worker", default=None, help=argparse.SUPPRESS)
    return ap.parse_args(argv)


# ------------------------------------------------------------------------------------------------------------------------
# self-launch: N ranks as fresh child processes (nothing in this parent has touched a GPU: torch is not even imported)
# ------------------------------------------------------------------------------------------------------------------------
def launch_ranks(n):
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    out, _ = procs[0].communicate()
    rcs = [procs[0].returncode]
    deadline = time.time() + 120
    for p in procs[1:]:
        try:
            rcs.append(p.wait(timeout=max(1.0, deadline - time.time())))
        except subprocess.TimeoutExpired:       # rank 0 is gone; a rank stuck in a collective is ended by its own PID
            p.kill()
            rcs.append(p.wait())
    sys.stdout.write(out.decode())
    sys.stdout.flush()
    bad = [rc for rc in rcs if rc != 0]
    return bad[0] if bad else 0


# ------------------------------------------------------------------------------------------------------------------------
# CPU baseline workers (the oracle is the CHECKER; here it is only the reported CPU baseline)
# ------------------------------------------------------------------------------------------------------------------------
def usable_cores():
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:                                             # cgroup v2 quota of the container / box share
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(int(q) / int(p))))
    except Exception:   # noqa: BLE001
        pass
    return max(1, n)


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except Exception:   # noqa: BLE001
        pass
    return "unknown"


def cpu_worker(spec):
    """child process of the all-cores leg: builds the oracle's preprocessing, says 'ready', waits for 'go', samples its row
    range of the job (disjoint i-ranges: row i depends only on (seed, i)), prints the seconds that took."""
    workload, mode, begin, count, m_total = spec.split(",")
    begin, count, m_total = int(begin), int(count), int(m_total)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle
    import ugs_workloads as wl
    ei, ptr, m, k = wl.workload(workload)
    G = len(ptr) - 1
    if G == 1:
        P = oracle.Preproc(ei, int(ptr[1]), k)
        run = lambda: P.sample(m_total, k, EDGE_MODE[mode], 0, 42, begin, begin + count)           # noqa: E731
    else:                                            # batches: the worker takes `count` samples per graph starting at sample `begin`
        cache = oracle.Cache()
        oracle.sample_batch(ei, ptr, 1, k, mode, 42, cache=cache)
        run = lambda: oracle.sample_batch(ei, ptr, count, k, mode, 42 + begin, cache=cache)  # noqa: E731
    print("ready", flush=True)
    sys.stdin.readline()
    t = time.perf_counter()
    run()
    print(f"done {time.perf_counter() - t:.6f}", flush=True)


def cpu_all_cores(workload, mode, m_total, G, rows_per_worker):
    cores = usable_cores()
    procs = []
    for w in range(cores):
        spec = f"{workload},{mode},{w * rows_per_worker},{rows_per_worker},{m_total}"
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), "--cpu-worker", spec], stdin=subprocess.PIPE,
                                      stdout=subprocess.PIPE, text=True))
    try:
        for p in procs:
            if p.stdout.readline().strip() != "ready":
                raise RuntimeError("CPU worker failed to start")
        t = time.perf_counter()
        for p in procs:
            p.stdin.write("go\n")
            p.stdin.flush()
        secs = [float(p.stdout.readline().split()[1]) for p in procs]
        wall = time.perf_counter() - t
    finally:
        for p in procs:
            try:
                p.stdin.close()
            except Exception:   # noqa: BLE001
                pass
            p.wait()
    rows = cores * rows_per_worker * (G if G > 1 else 1)
    return {"value": round(rows / wall, 1), "unit": "k-subgraphs/s", "cores": cores, "kind": "port", "cpu_model": cpu_model(),
            "sample": f"{cores} processes x {rows_per_worker} {'samples per graph' if G > 1 else 'rows'} of the same job (disjoint sample-index ranges), "
                      f"wall {wall:.2f}s from a common start, slowest worker {max(secs):.2f}s; preprocessing excluded",
            "per_core": round(rows / wall / cores, 1)}


def csr_degrees(ei, n_total):
    import numpy as np
    ok = (ei[0] >= 0) & (ei[1] >= 0) & (ei[0] < n_total) & (ei[1] < n_total)
    return np.bincount(ei[0][ok], minlength=n_total) + np.bincount(ei[1][ok], minlength=n_total)


def split_algorithmic_bytes(nodes, edge_ptr, k, deg):
    """SURVEY.md 8(d) per-sample algorithmic bytes, split by the kernel that has to move them:
    walk = 16 [alias row + root vertex] + sum_{v in S}(16 + 4 deg v) + sum_{v in S[:k-1]} 4 deg v + 8k + 8;
    fill = 4 Es + 24 Es.  walk + fill == the 8(d) figure."""
    import numpy as np
    valid = nodes >= 0
    d = np.where(valid, deg[np.where(valid, nodes, 0)], 0).astype(np.int64)
    es = np.diff(edge_ptr).astype(np.int64)
    walk = 16 + 16 * valid.sum(1) + 4 * d.sum(1) + 4 * d[:, : max(k - 1, 0)].sum(1) + 8 * k + 8
    fill = 28 * es
    return float(walk.mean()), float(fill.mean())


class Job:
    """One sharded sampling job on this rank: buffers, the step function and (N > 1) its part of the collation.

    Stream plan of a step (N > 1).  Everything a rank computes runs on its MAIN stream, back to back: walk, scan, fill of batch i;
    on the destination the unpack of batch i-1 (whose messages arrived while batch i was being sampled); the pack of batch i.
    Only the exchange itself (RCCL gather) runs on a side stream, behind the pack.  Round 2 ran pack and unpack on the side stream
    BESIDE the next walk: the walk kernels are persistent grids that hold their share of every CU to the end, so the collation got
    what was left (a fifth of the machine) and took up to a whole step -- the steady state then depended on how the two streams
    happened to interleave (rank-0 emulation: 0.76 ... 1.03 ms per step for neighbouring shard sizes).  Serial on one stream the
    same kernels take 0.2 ms at full rate, and the step time is the sum of its parts: what the split calibration needs.

    Two steps in flight (N > 1).  A rank's shard is small (125 k rows of the 1 M: 24 walks per wave), so a step alone on the GPU
    pays its launch ramp, its tail and four launch gaps around kernels that do not fill the chip (scan, fill, pack): a sixth of the
    step.  Consecutive steps therefore alternate between TWO streams, each with its own buffer set and its own plan scratch
    (Plan.twin(): same device arrays): step i+1's walk starts while step i's tail and small kernels run.  The collation chain
    stays ordered by events: unpack(i-1) and pack(i) on step i's stream behind exchange(i-1), exchange(i) behind pack(i)."""

    def __init__(self, torch, dist, ud, plan, args, G, m_total, k, rank, world, dev, node_bound, n_cols, use_collate, weights=None):
        self.torch, self.plan, self.args, self.k, self.m_total = torch, plan, args, k, m_total
        self.total_rows = G * m_total
        self.row_off = ud.shard_offsets(self.total_rows, world, weights)      # the same list on every rank
        self.row_begin, self.row_count = self.row_off[rank], self.row_off[rank + 1] - self.row_off[rank]
        self.use_collate = use_collate
        self.nsets = 2 if ((use_collate or getattr(args, "two_steps", False)) and not args.one_stream) else 1     # steps alternate between two streams / buffer sets / plan scratches
        rc = self.row_count
        self.nodes = [torch.empty((rc, k), dtype=torch.int64, device=dev) for _ in range(self.nsets)]
        self.eptr = [torch.empty((rc + 1,), dtype=torch.int64, device=dev) for _ in range(self.nsets)]
        # edge capacity from one synchronous probe step (+5%); identical on every rank
        _, _, tot = plan.walk(m_total, args.mode, 41, self.row_begin, rc, out=(self.nodes[0], self.eptr[0]), sync=True)
        cap_t = torch.tensor([int(tot * 1.05) + 4096], dtype=torch.int64, device=dev)
        if world > 1:
            dist.all_reduce(cap_t, op=dist.ReduceOp.MAX)
        self.cap = int(cap_t.item())
        self.eidx = [torch.empty((2, self.cap), dtype=torch.int64, device=dev) for _ in range(self.nsets)]
        self.esrc = [torch.empty((self.cap,), dtype=torch.int64, device=dev) for _ in range(self.nsets)]
        self.main = torch.cuda.current_stream()
        self.streams = [self.main] + [torch.cuda.Stream(device=dev) for _ in range(self.nsets - 1)]
        self.plans = [plan] + [plan.twin() for _ in range(self.nsets - 1)]
        if use_collate:
            self.side = torch.cuda.Stream(device=dev)
            self.collator = ud.Collator(self.total_rows, k, args.mode, node_bound, max(node_bound, m_total * k), n_cols, self.cap, dev, dst=0,
                                        row_off=self.row_off)
            self.ev_packed, self.ev_exchanged = torch.cuda.Event(), torch.cuda.Event()
            self.in_flight = False             # a batch has been packed and handed to the exchange, not unpacked yet
            self.cev = []                      # (start, end) events of this rank's collation work (unpack + pack) on the main stream
        self.totals = None

    def sample(self, i):
        """walk + scan + fill of step i (asynchronous; Plan.step) on the stream of buffer set i % nsets"""
        b = i % self.nsets
        with self.torch.cuda.stream(self.streams[b]):
            self.plans[b].step(self.m_total, self.args.mode, 42 + i, self.row_begin, self.row_count, out=(self.nodes[b], self.eptr[b], self.eidx[b], self.esrc[b]))
            if self.totals is not None:
                self.totals[i] = self.eptr[b][-1]

    def _unpack_previous(self, stream):
        """the batch whose exchange is in flight: wait for its messages (stream-side), unpack it on the destination"""
        res = None
        if self.in_flight:
            stream.wait_event(self.ev_exchanged)                 # also: this rank's message buffer is free again
            with self.torch.cuda.stream(stream):
                res = self.collator.unpack()
            self.in_flight = False
        return res

    def collate_step(self, i, timed, exchange=True):
        """this rank's part of the one exchange step, after sample(i): unpack batch i-1 (destination), pack batch i, start its
        exchange.  Returns the collated batch i-1 on the destination.  exchange=False (calibration): the same work without the
        collective -- the destination unpacks the messages already in its inbox."""
        torch, b = self.torch, i % self.nsets
        st = self.streams[b]
        if timed:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(st)
        res = self._unpack_previous(st)
        with torch.cuda.stream(st):
            self.collator.pack((self.nodes[b], self.eidx[b], self.eptr[b], self.esrc[b]))
        self.ev_packed.record(st)
        if timed:
            e1.record(st)
            self.cev.append((e0, e1))
        if exchange:
            with torch.cuda.stream(self.side):
                self.side.wait_event(self.ev_packed)
                self.collator.exchange()
                self.ev_exchanged.record(self.side)
        else:
            self.ev_exchanged.record(st)
        self.in_flight = True
        return res

    def run_steps(self, first, count, timed=False, exchange=True):
        """`count` complete steps: every batch sampled AND (multi-GPU) collated inside the call; returns the last collated batch"""
        res = None
        for i in range(first, first + count):
            self.sample(i)
            if self.use_collate:
                self.collate_step(i, timed, exchange)
        if self.use_collate and count > 0:
            res = self._unpack_previous(self.main)               # the last batch: its exchange is exposed, as in any pipeline's drain
            self.side.synchronize()
        for st in self.streams[1:]:                              # the caller's stream sees everything that was enqueued
            self.main.wait_stream(st)
        return res

    def local_ms_per_step(self, steps=48):
        """calibration: this rank's OWN work per step in steady state (sampling; pack; on the destination the unpack of a whole
        batch beside the next step's sampling) with no exchange, so that no rank's time contains another rank's"""
        torch = self.torch
        self.run_steps(2000, 2, exchange=False)
        torch.cuda.synchronize()
        t = time.perf_counter()
        self.run_steps(2002, steps, exchange=False)
        torch.cuda.synchronize()
        return (time.perf_counter() - t) / steps * 1e3


def timed_run(torch, dist, job, steps, warmup, world, dev):
    job.totals = torch.zeros((steps + warmup + 1,), dtype=torch.int64, device=dev)
    job.run_steps(0, warmup)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t_start = time.perf_counter()
    job.run_steps(warmup, steps, timed=True)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t_start
    el_t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(el_t, op=dist.ReduceOp.MAX)
    tmax = int(job.totals.max().item())
    if tmax > job.cap:
        raise SystemExit(f"edge capacity {job.cap} too small for {tmax}: result invalid")
    cms = None
    if job.use_collate and job.cev:
        cms = sum(a.elapsed_time(b) for a, b in job.cev) / len(job.cev)
    return float(el_t.item()), cms


def main():
    args = parse_args()
    if args.cpu_worker:
        return cpu_worker(args.cpu_worker)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args.gpus))

    # Libraries (RCCL prints a version banner) must not pollute stdout: fd 1 is pointed at stderr for the whole run and the
    # single JSON line goes to the saved, real stdout at the end.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    def emit(line):
        os.write(real_stdout, (line + "\n").encode())

    import numpy as np
    import torch
    import torch.distributed as dist

    import ugs_sampler
    import ugs_workloads as wl
    from ugs_sampler import distributed as ud

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.rehearse:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1 or args.force_collate:
        if "MASTER_ADDR" not in os.environ:
            os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", os.environ.get("MASTER_PORT", "29517")
        if args.rehearse:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    # ---- workload + plan (host preprocessing and upload are NOT timed: warm-cache state) --------------------------
    t0 = time.time()
    ei, ptr, m, k = wl.workload(args.workload)
    G = len(ptr) - 1
    ei_t, ptr_t = torch.from_numpy(ei), torch.from_numpy(ptr)
    t1 = time.time()
    plan = ugs_sampler.Plan.from_batch(ei_t, ptr_t, k, device=dev)
    t2 = time.time()
    info = plan.info()
    if rank == 0:
        log(f"[bench] workload {args.workload}: G={G} cols={ei.shape[1]} k={k} batch={G * m}; generate {t1 - t0:.1f}s, "
            f"preprocess+upload {t2 - t1:.1f}s, plan {info['device_bytes'] / 1e6:.0f} MB in HBM, tier {info['tier']}, {world} rank(s), {args.scaling} scaling")
    node_bound = int(ptr[-1])
    use_collate = world > 1 or args.force_collate
    m_total = m * world if args.scaling == "weak" else m
    my_share = 100
    if use_collate:
        # the destination's walk kernels leave room on every CU for the unpack + RCCL kernels of the previous batch; the other ranks
        # only pack and send (copy kernels of a few microseconds) and keep the whole CU
        my_share = args.walk_share if rank == 0 else args.walk_share_others
        plan.set_walk_share(my_share)
    plan.set_timing(False)
    # ---- the split of the rows over the ranks (N > 1): equal, given, or measured in the warm-up ---------------------------------
    weights, calibration = None, None
    if world > 1 and args.dst_rows != "equal":
        if args.dst_rows != "auto":
            weights = [float(args.dst_rows)] + [1.0] * (world - 1)
        else:
            # Rank 0 also unpacks the whole batch, so with equal shards it is the critical path of every step.  Each round: one real
            # step (fills the destination's inbox), then every rank times its own steady-state work without the exchange; one
            # all-gather of (rows, ms); next weights = rows per millisecond.  A rank's time is rows * s + f with a fixed part f
            # (the destination's unpack), so the proportional update is repeated (three rounds): it contracts towards equal times.
            calibration = []
            tried = []                                                    # (weights a round ran with, the slowest rank's ms in that round)
            for _ in range(3):
                cj = Job(torch, dist, ud, plan, args, G, m_total, k, rank, world, dev, node_bound, ei.shape[1], True, weights)
                cj.run_steps(0, 1)
                torch.cuda.synchronize()
                mine = torch.tensor([float(cj.row_count), cj.local_ms_per_step()], dtype=torch.float64, device=dev)
                every = torch.empty((world, 2), dtype=torch.float64, device=dev)
                dist.all_gather_into_tensor(every, mine.reshape(1, 2))
                every = every.cpu().tolist()
                calibration.append({"rows": [int(r_) for r_, _ in every], "local_ms_per_step": [round(t_, 4) for _, t_ in every]})
                tried.append((weights, max(t_ for _, t_ in every)))
                weights = [max(r_, 1.0) / max(t_, 1e-6) for r_, t_ in every]
                mid = sorted(weights)[len(weights) // 2]                  # one stalled measurement must not starve (or flood) a rank
                weights = [min(max(w_, 0.5 * mid), 1.5 * mid) for w_ in weights]
                del cj
            # Guard.  The update contracts towards equal times only while a rank's time is a smooth function of its rows; with two
            # steps in flight it is not (+-4 % by how the two launches interlock), and one jittery round can send the split the
            # wrong way.  While every round was faster than the one before, the next (derived, untried) split continues the trend
            # and is taken; otherwise the job runs with the best split that was actually MEASURED.  Every rank sees the same
            # all-gathered numbers, so every rank takes the same decision.
            slowest = [t_ for _, t_ in tried]
            monotone = all(b_ < a_ for a_, b_ in zip(slowest, slowest[1:]))
            if not monotone:
                weights = min(tried, key=lambda wt: wt[1])[0]
            calibration.append({"slowest_rank_ms": [round(t_, 4) for t_ in slowest], "monotone": monotone,
                                "split": "derived from the last round" if monotone else "best measured round"})
    job = Job(torch, dist, ud, plan, args, G, m_total, k, rank, world, dev, node_bound, ei.shape[1], use_collate, weights)
    total_rows, row_begin, row_count = job.total_rows, job.row_begin, job.row_count

    job.run_steps(0, 1)                                   # first touch of every buffer outside the event-timed region
    torch.cuda.synchronize()
    plan.set_timing(True)
    elapsed, collate_ms = timed_run(torch, dist, job, args.steps, args.warmup, world, dev)
    timing = plan.get_timing()
    plan.set_timing(False)
    per_rank = None
    if use_collate:                                       # every rank's rows and kernel times, for rank 0's line (a collective: before the ranks part)
        tm = lambda key: timing[key][0] / max(timing[key][1], 1)     # noqa: E731
        mine = torch.tensor([float(row_count), tm("walk"), tm("scan"), tm("fill"), float(collate_ms or 0.0), float(my_share)], dtype=torch.float64, device=dev)
        every = torch.empty((world, 6), dtype=torch.float64, device=dev)
        if world > 1:
            dist.all_gather_into_tensor(every, mine.reshape(1, 6))
        else:
            every.copy_(mine.reshape(1, 6))
        per_rank = [{"rank": r_, "rows": int(v[0]), "walk_ms": round(v[1], 4), "scan_ms": round(v[2], 4), "fill_ms": round(v[3], 4),
                     "collate_ms": round(v[4], 4), "walk_share": int(v[5])} for r_, v in enumerate(every.cpu().tolist())]
        if job.collator.overflowed():                     # lazy capacity check of the collation (one read-back, after the timed region)
            raise SystemExit("a step's edge total exceeded the collation capacity: result invalid")

    extras = {}
    if world > 1 and args.scaling == "strong" and not args.no_extras:
        # the other scaling mode, short: every rank adds a whole batch (N*B rows per step collated on rank 0)
        wjob = Job(torch, dist, ud, plan, args, G, m * world, k, rank, world, dev, node_bound, ei.shape[1], True, weights)
        wsteps = max(3, args.steps // 2)
        wel, wcms = timed_run(torch, dist, wjob, wsteps, 2, world, dev)
        extras["weak_scaling"] = {"value": round(wjob.total_rows * wsteps / wel, 1), "ms_per_step": round(wel / wsteps * 1e3, 4), "steps": wsteps,
                                  "global_rows": wjob.total_rows, "rows_per_gpu": wjob.row_count, "collate_ms_per_step": round(wcms, 4) if wcms else None}
        del wjob

    # the collated batch of one more step against rank 0's own rows (placement check of the exchange step): a collective, so
    # every rank takes part before the ranks part ways
    if use_collate:
        job.totals = None
        last = job.run_steps(1000, 1)
        torch.cuda.synchronize()
        if rank == 0:
            b = 1000 % job.nsets
            c_nodes, c_eidx, c_eptr, c_esrc = last
            rb, rc_ = job.row_begin, job.row_count
            assert int(c_eptr[-1]) >= int(job.eptr[b][-1]) and bool((c_eptr[1:] >= c_eptr[:-1]).all()), "collated edge_ptr is not a scan"
            assert torch.equal(c_nodes[rb:rb + rc_], job.nodes[b]), "collated nodes differ"
            assert torch.equal(c_eptr[rb:rb + rc_ + 1] - c_eptr[rb], job.eptr[b]), "collated edge_ptr differs"
            e0, e1 = int(c_eptr[rb]), int(c_eptr[rb + rc_])
            assert torch.equal(c_eidx[:, e0:e1], job.eidx[b][:, : e1 - e0]) and torch.equal(c_esrc[e0:e1], job.esrc[b][: e1 - e0]), "collated edges differ"
            if world > 1:      # and the rows of the OTHER ranks: the same rows sampled here, unsharded (row i depends only on (seed, i))
                chk_n = min(20_000, job.total_rows - rc_)
                o_nodes, o_eidx, o_eptr, o_esrc = plan.sample_rows(m_total, args.mode, 42 + 1000, rb + rc_, chk_n)
                assert torch.equal(c_nodes[rb + rc_:rb + rc_ + chk_n], o_nodes), "collated rows of another rank differ from the unsharded rows"
                f0 = int(c_eptr[rb + rc_])
                assert torch.equal(c_eptr[rb + rc_:rb + rc_ + chk_n + 1] - f0, o_eptr) and torch.equal(c_eidx[:, f0:f0 + o_eidx.size(1)], o_eidx) \
                    and torch.equal(c_esrc[f0:f0 + o_esrc.numel()], o_esrc), "collated edges of another rank differ from the unsharded rows"

    if rank != 0:
        dist.barrier()
        dist.destroy_process_group()
        return

    value = total_rows * args.steps / elapsed
    ms_per_step = elapsed / args.steps * 1e3
    # timing["walk"] etc. include the warm-up steps of timed_run: per-launch means
    walk_ms = timing["walk"][0] / max(timing["walk"][1], 1)
    scan_ms = timing["scan"][0] / max(timing["scan"][1], 1)
    fill_ms = timing["fill"][0] / max(timing["fill"][1], 1)
    launch = plan.last_launch()

    # ---- a reference step (seed 42): algorithmic bytes, and parity of ALL FOUR tensors of its first rows against the oracle ----
    nodes_buf, eptr_buf, eidx_buf, esrc_buf = job.nodes[0], job.eptr[0], job.eidx[0], job.esrc[0]
    _, _, tot42 = plan.walk(m_total, args.mode, 42, row_begin, row_count, out=(nodes_buf, eptr_buf), sync=True)
    plan.fill(m_total, nodes_buf, eptr_buf, None, args.mode, row_begin, out=(eidx_buf, esrc_buf))
    torch.cuda.synchronize()
    nodes_h = nodes_buf.cpu().numpy()
    eptr_h = eptr_buf.cpu().numpy()
    deg = csr_degrees(ei, node_bound)
    walk_bytes, fill_bytes = split_algorithmic_bytes(nodes_h, eptr_h, k, deg)
    unit_bytes = walk_bytes + fill_bytes
    achieved = walk_bytes * row_count / (walk_ms * 1e-3) / 1e9 if walk_ms > 0 else 0.0
    gpu_ms = walk_ms + scan_ms + fill_ms
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")   # written from separate rocprofv3 --pmc passes of this command
    if os.path.exists(tpath) and world == 1:
        try:
            with open(tpath) as f:
                traffic = json.load(f).get(args.workload, {}).get("walk_kernel_hbm_bytes_per_launch")
        except Exception:   # noqa: BLE001
            traffic = None
    roofline = {"bound": "hbm", "kernel": launch["kernel"], "achieved": round(achieved, 2), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBPS, 5), "traffic": traffic,
                "algorithmic_bytes_per_unit": round(walk_bytes, 1), "units_per_launch": row_count,
                "kernel_ms": round(walk_ms, 4), "grid": launch["grid"], "block": launch["block"], "lds_bytes_per_block": launch["lds_bytes"],
                "path": {"algorithmic_bytes_per_unit": round(unit_bytes, 1), "gpu_ms_per_step": round(gpu_ms, 4),
                         "achieved": round(unit_bytes * row_count / (gpu_ms * 1e-3) / 1e9, 2) if gpu_ms > 0 else 0.0,
                         "fill_kernel_ms": round(fill_ms, 4), "scan_ms": round(scan_ms, 4)}}

    cpu_baseline = None
    cpu_all = None
    parity_rows = 0
    if not args.no_cpu_baseline:
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import oracle   # the checker, used here only as the reported CPU baseline and for a parity check
        n_cpu = args.cpu_sample if args.cpu_sample > 0 else (100_000 if args.workload.startswith("c5") else min(total_rows, 400_000))
        if G == 1:
            tp = time.time()
            P = oracle.Preproc(ei, int(ptr[1]), k)
            tp = time.time() - tp
            n_cpu = min(n_cpu, row_count)
            tc = time.perf_counter()
            o_nodes, o_eidx, o_eptr, o_esrc = P.sample(m_total, k, EDGE_MODE[args.mode], 0, 42, row_begin, row_begin + n_cpu)
            tc = time.perf_counter() - tc
            sample_desc = f"rows [{row_begin},{row_begin + n_cpu}) of the same {m_total}-row job on the same graph, seed 42; preprocessing ({tp:.1f}s) excluded like the warm-cache GPU plan"
            parity_rows = n_cpu
            ne = int(o_eptr[-1])
            assert int(eptr_h[n_cpu]) == ne and np.array_equal(o_nodes, nodes_h[:n_cpu]) and np.array_equal(o_eptr, eptr_h[: n_cpu + 1]), "GPU rows differ from the CPU oracle"
            assert np.array_equal(o_eidx, eidx_buf[:, :ne].cpu().numpy()) and np.array_equal(o_esrc, esrc_buf[:ne].cpu().numpy()), "GPU edges differ from the CPU oracle"
            P.close()
        else:
            cache = oracle.Cache()
            oracle.sample_batch(ei, ptr, 1, k, args.mode, 42, cache=cache)          # warm the LRU (preprocessing excluded)
            m_cpu = max(1, min(m_total, n_cpu // G))
            tc = time.perf_counter()
            o = oracle.sample_batch(ei, ptr, m_cpu, k, args.mode, 42, cache=cache)
            tc = time.perf_counter() - tc
            n_cpu = m_cpu * G
            sample_desc = f"the same {G}-graph batch with m_per_graph={m_cpu} ({n_cpu} rows), seed 42, warm preprocessing LRU"
            if m_cpu == m_total and world == 1:
                parity_rows = n_cpu
                ne = int(o[2][-1])
                assert np.array_equal(o[0], nodes_h) and np.array_equal(o[2], eptr_h), "GPU rows differ from the CPU oracle"
                assert np.array_equal(o[1], eidx_buf[:, :ne].cpu().numpy()) and np.array_equal(o[4], esrc_buf[:ne].cpu().numpy()), "GPU edges differ from the CPU oracle"
        cpu_baseline = {"value": round(n_cpu / tc, 1), "unit": "k-subgraphs/s", "cores": 1, "kind": "port",
                        "sample": sample_desc + "; oracle/ugs_oracle.c (plain-C restatement of the reference algorithm), 1 thread", "cpu_model": cpu_model()}
        cpu_port = None
        if world == 1 and not args.no_cpu_reference:          # the reference C++ itself, when its prebuilt copy travelled (oracle/_ref)
            try:
                cref = reference_on_headline(ei_t, ptr_t, m_total, k, args.mode, min(m_total, 20_000 if G == 1 else m_total), G)
                if cref is not None:
                    cpu_port, cpu_baseline = cpu_baseline, cref
            except Exception as e:   # noqa: BLE001
                log(f"[bench] reference C++ on the headline workload failed: {e}")
        if world == 1:
            try:
                per = max(1, min(20_000, m_total // usable_cores())) if G == 1 else max(1, min(m_total, 100_000 // G))
                cpu_all = cpu_all_cores(args.workload, args.mode, m_total, G, per)
            except Exception as e:   # noqa: BLE001
                cpu_all = {"error": str(e)[:200]}

    out = {"metric": "k_subgraphs_sampled_per_sec", "value": round(value, 1), "unit": "k-subgraphs/s", "n_gpus": world,
           "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
           "scaling": args.scaling, "vs_baseline": None, "dtype": "int64", "data": "synthetic",
           "config": {"workload": args.workload, "graphs": G, "columns": int(ei.shape[1]), "k": k, "rows_per_gpu": row_count,
                      "global_rows": total_rows, "mode": args.mode, "sharding": f"rows{world}" if world > 1 else "none",
                      "collate": "gather to rank 0 over RCCL every step (fixed-size narrowed messages, device-side offsets, no host round trip), "
                                 f"the gather overlaps the next step's sampling, rank 0 unpacks one step behind (walk kernels on {args.walk_share}% of each "
                                 f"CU on the destination so that RCCL's kernels find wave slots, {args.walk_share_others}% elsewhere); rows split {args.dst_rows}"
                                 if use_collate else "none (single GPU)"},
           "collate_ms_per_step": round(collate_ms, 4) if collate_ms is not None else None,
           "per_rank": per_rank, "split_calibration": calibration,
           "roofline": roofline, "cpu_baseline": cpu_baseline, "cpu_baseline_port": cpu_port if cpu_baseline else None,
           "cpu_baseline_all_cores": cpu_all, "parity_checked_rows": parity_rows,
           "parity_checked_tensors": ["nodes", "edge_ptr", "edge_index", "edge_src"] if parity_rows else []}
    if extras:
        out["extras"] = extras

    # ---- secondary measurements (single GPU only, quick) ------------------------------------------------------------------
    if not args.no_extras and world == 1:
        try:
            out["drop_in_call"] = bench_drop_in(args.workload, ugs_sampler, ei_t, ptr_t, m, k, args.mode, dev, reps=3 if ei.shape[1] > 5_000_000 else 20)
        except Exception as e:   # noqa: BLE001
            out["drop_in_call"] = {"error": str(e)[:200]}
        extras_w = {}
        for name in ("c2_mutag_b1024", "c3_proteins_b8192", "c4_qm9_b65536"):
            if name == args.workload:
                continue
            try:
                extras_w[name] = bench_small(name, ugs_sampler, wl, torch, dev)
            except Exception as e:   # noqa: BLE001
                extras_w[name] = {"error": str(e)}
        out["other_workloads"] = extras_w
        try:
            out["epsilon_uniform_sampler"] = bench_epsilon(wl, torch)
        except Exception as e:   # noqa: BLE001
            out["epsilon_uniform_sampler"] = {"error": str(e)[:200]}
        try:
            pr = port_vs_reference_on_er_proxy(wl, torch)
            if pr is not None:
                out["cpu_baseline_calibration"] = pr
        except Exception as e:   # noqa: BLE001
            out["cpu_baseline_calibration"] = {"error": str(e)[:200]}
    if world == 1:
        hv = (out.get("drop_in_call") or {}).get("host_visible_subgraphs_per_s")
        if hv:      # SURVEY.md 8(d)'s metric -- wall time of the sample_batch call with host-visible outputs -- beside `value` (inputs and outputs resident in HBM)
            out["value_host_visible"] = hv
        try:
            out["summary"] = summary_block(out, args.workload)      # LAST key: the driver's record keeps the tail of the line
        except Exception as e:   # noqa: BLE001
            out["summary"] = {"error": str(e)[:100]}
    sys.stdout.flush()
    emit(json.dumps(out))
    if world > 1 or args.force_collate:
        dist.barrier()
        dist.destroy_process_group()


def bench_drop_in(name, ugs_sampler, ei_t, ptr_t, m, k, mode, dev, reps):
    """The reference's own call, end to end on the headline workload: ugs_sampler.sample_batch(edge_index, ptr, ...) with host
    tensors in -- (a) pinned host tensors out (SURVEY.md 8(d) primary definition: outputs host-visible), (b) device tensors out."""
    import torch
    rows = (ptr_t.numel() - 1) * m
    streamed = rows >= ugs_sampler._STREAM_MIN_ROWS and not os.environ.get("UGS_NO_STREAMED_CALL")
    two_phase_ms = None
    if streamed:        # the two-phase form of the same call (walks, then fill, then copy-out), for the record
        os.environ["UGS_NO_STREAMED_CALL"] = "1"
        ugs_sampler.sample_batch(ei_t, ptr_t, m, k, mode=mode, seed=42)
        t = time.perf_counter()
        for r in range(reps):
            ugs_sampler.sample_batch(ei_t, ptr_t, m, k, mode=mode, seed=42 + r)
        two_phase_ms = round((time.perf_counter() - t) / reps * 1e3, 3)
        del os.environ["UGS_NO_STREAMED_CALL"]
    for r in range(2):  # (a shape's first call sizes the edge buffers of the later ones: ugs_sampler._sample_batch_streamed)
        ugs_sampler.sample_batch(ei_t, ptr_t, m, k, mode=mode, seed=40 + r)
    t = time.perf_counter()
    for r in range(reps):
        ugs_sampler.sample_batch(ei_t, ptr_t, m, k, mode=mode, seed=42 + r)
    dt_host = (time.perf_counter() - t) / reps
    ugs_sampler.sample_batch(ei_t, ptr_t, m, k, mode=mode, seed=42, device=dev)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for r in range(reps):
        o = ugs_sampler.sample_batch(ei_t, ptr_t, m, k, mode=mode, seed=42 + r, device=dev)
    torch.cuda.synchronize()
    dt_dev = (time.perf_counter() - t) / reps
    del o
    return {"workload": name, "rows": rows, "host_visible_ms": round(dt_host * 1e3, 3), "host_visible_subgraphs_per_s": round(rows / dt_host, 1),
            "device_out_ms": round(dt_dev * 1e3, 3), "device_out_subgraphs_per_s": round(rows / dt_dev, 1), "reps": reps,
            "host_visible_streamed": bool(streamed), "host_visible_two_phase_ms": two_phase_ms,
            "note": "host tensors in every call (the reference's interface): per call the library hashes the batch's bytes (a batch seen before is matched as "
                    "a whole; the per-graph LRU is touched as the general path would), samples, and copies out; calls of >= 262144 rows are streamed "
                    "(row chunks copied out beside the walks, walks begun while the hash runs: ugs_sample_batch_stream)"}


def small_traffic(name):
    """HBM bytes per walk launch of a TU-shaped workload from the PMC passes kept in profiles/pmc_traffic.json (None if not profiled)"""
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
            return json.load(f).get(name, {}).get("walk_kernel_hbm_bytes_per_launch")
    except Exception:   # noqa: BLE001
        return None


def summary_block(out, workload):
    """<= 600 characters at the END of the JSON line: what SURVEY.md 8(d) asks for -- the host-visible rate of the sample_batch call
    (M k-subgraphs/s: repeated batch / new combination of known graphs / all graphs new to the LRU) beside the reference C++ on the
    same box (k/s, one core) and the ratio, for the PROTEINS- and QM9-shaped batches and the headline job; and what binds the walk
    kernel (instructions per walk, busy share of the vector issue port: tracked PMC passes)."""
    def m3(x):
        return None if x is None else float(f"{x:.3g}")
    s = {}
    for key, name in (("c3", "c3_proteins_b8192"), ("c4", "c4_qm9_b65536")):
        w = out.get("other_workloads", {}).get(name) or {}
        if "rows" not in w:
            continue
        rows, ref = w["rows"], (w.get("cpu_baseline") or {}).get("value")
        cold = (w.get("drop_in_call_cold") or {}).get("device_ms")
        e = {"warm": m3(rows / w["drop_in_call_ms"] / 1e3), "new": m3(rows / w["drop_in_call_shuffled_batch_ms"] / 1e3),
             "cold": m3(rows / cold / 1e3) if cold else None, "dev": m3(w["device_resident_subgraphs_per_s"] / 1e6), "ref_k": m3(ref / 1e3) if ref else None}
        if ref and cold:
            e["x_warm"], e["x_cold"] = m3(rows / w["drop_in_call_ms"] * 1e3 / ref), m3(rows / cold * 1e3 / ref)
        s[key] = e
    d = out.get("drop_in_call") or {}
    ref = (out.get("cpu_baseline") or {}).get("value")
    hv = d.get("host_visible_subgraphs_per_s")
    s[workload[:2]] = {"dev": m3(out["value"] / 1e6), "hv": m3(hv / 1e6) if hv else None, "ref_k": m3(ref / 1e3) if ref else None,
                       "x_hv": m3(hv / ref) if hv and ref else None}
    tp = (out.get("drop_in_call") or {}).get("host_visible_two_phase_ms")
    if tp:                                  # the same call without the streamed copy-out (walks, then fill, then copy): M/s
        s[workload[:2]]["hv_2ph"] = m3((out["drop_in_call"]["rows"] / (tp * 1e-3)) / 1e6)
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
            iss = json.load(f).get(workload, {}).get("issue")
        if iss:
            s["issue"] = {"inst": iss.get("instructions_per_walk"), "valu": iss.get("valu_per_walk"), "salu": iss.get("salu_per_walk"),
                          "valu_busy": iss.get("valu_busy_share_of_simd_quad_cycles")}
    except Exception:   # noqa: BLE001
        pass
    s["units"] = "M/s; ref_k k/s 1 core"
    return s


def bench_small(name, ugs_sampler, wl, torch, dev, reps=50):
    """TU-shaped configuration: (a) device-resident plan path, outputs in HBM (per-repetition HIP-event times: median and max);
    (b) the drop-in host call ugs_sampler.sample_batch(...) end to end (slice + hash + LRU lookups, kernels, D2H into pinned tensors)."""
    import numpy as np
    ei, ptr, m, k = wl.workload(name)
    G = len(ptr) - 1
    rows = G * m
    ei_t, ptr_t = torch.from_numpy(ei), torch.from_numpy(ptr)
    plan = ugs_sampler.Plan.from_batch(ei_t, ptr_t, k, device=dev)
    nodes = torch.empty((rows, k), dtype=torch.int64, device=dev)
    eptr = torch.empty((rows + 1,), dtype=torch.int64, device=dev)
    _, _, tot = plan.walk(m, "sample", 42, 0, rows, out=(nodes, eptr), sync=True)
    cap = int(tot * 1.2) + 1024
    eidx = torch.empty((2, cap), dtype=torch.int64, device=dev)
    esrc = torch.empty((cap,), dtype=torch.int64, device=dev)
    for i in range(5):
        plan.step(m, "sample", 42 + i, 0, rows, out=(nodes, eptr, eidx, esrc))
    torch.cuda.synchronize()
    evs = []
    t = time.perf_counter()
    for i in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        plan.step(m, "sample", 42 + i, 0, rows, out=(nodes, eptr, eidx, esrc))
        b.record()
        evs.append((a, b))
    torch.cuda.synchronize()
    dt_wall = (time.perf_counter() - t) / reps
    # roofline block of this workload's dominant kernel: per-kernel HIP events of the library on the launch stream (a second loop,
    # so that the events do not sit inside the step times above), SURVEY.md 8(d) bytes measured on the step's own output
    plan.set_timing(True)
    for i in range(reps):
        plan.step(m, "sample", 42 + i, 0, rows, out=(nodes, eptr, eidx, esrc))
    torch.cuda.synchronize()
    tk = plan.get_timing()
    plan.set_timing(False)
    kms = {key: tk[key][0] / max(tk[key][1], 1) for key in ("walk", "scan", "fill")}
    launch = plan.last_launch()
    plan.walk(m, "sample", 42, 0, rows, out=(nodes, eptr), sync=True)
    wb, fb = split_algorithmic_bytes(nodes.cpu().numpy(), eptr.cpu().numpy(), k, csr_degrees(ei, int(ptr[-1])))
    ach = wb * rows / (kms["walk"] * 1e-3) / 1e9 if kms["walk"] > 0 else 0.0
    roofline = {"bound": "hbm", "kernel": launch["kernel"], "achieved": round(ach, 2), "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBPS, 5),
                "traffic": small_traffic(name), "algorithmic_bytes_per_unit": round(wb, 1), "units_per_launch": rows, "kernel_ms": round(kms["walk"], 4),
                "grid": launch["grid"], "block": launch["block"], "lds_bytes_per_block": launch["lds_bytes"],
                "path": {"algorithmic_bytes_per_unit": round(wb + fb, 1), "scan_ms": round(kms["scan"], 4), "fill_kernel_ms": round(kms["fill"], 4),
                         "gpu_ms_per_step": round(sum(kms.values()), 4)},
                "note": "graphs of this size are L2-resident: the kernel is latency- and launch-bound, the HBM fraction is reported for completeness"}
    per = sorted(a.elapsed_time(b) for a, b in evs)
    dt_dev = per[len(per) // 2] * 1e-3                                     # median repetition (a single stalled one does not move it)
    # the same step captured once as a HIP graph and replayed (Plan.graph_step)
    # EVERY replay is timed twice -- HIP events around it on the stream (device side) and the host clock around the launch call --
    # and the median, the maximum and the index of the maximum are reported: a bare mean over the loop hid a single stalled
    # replay (round 2: 1.77 ms "per replay" in the driver's run against 0.09 ms in every other run).
    dt_graph, graph_stats = None, None
    try:
        step = plan.graph_step(m, "sample", 0, rows, edge_capacity=cap)
        for i in range(5):
            step.launch(42 + i)
        torch.cuda.synchronize()
        gev, host_us = [], []
        t = time.perf_counter()
        for i in range(reps):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            th = time.perf_counter()
            step.launch(42 + i)
            host_us.append((time.perf_counter() - th) * 1e6)
            b.record()
            gev.append((a, b))
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t) / reps
        dev_ms = [a.elapsed_time(b) for a, b in gev]
        srt = sorted(dev_ms)
        dt_graph = srt[len(srt) // 2] * 1e-3
        graph_stats = {"median_ms": round(srt[len(srt) // 2], 4), "max_ms": round(srt[-1], 4), "index_of_max": int(dev_ms.index(srt[-1])),
                       "wall_mean_ms": round(wall * 1e3, 4), "host_launch_call_us_median": round(sorted(host_us)[len(host_us) // 2], 1),
                       "host_launch_call_us_max": round(max(host_us), 1), "index_of_host_max": int(host_us.index(max(host_us))), "replays": reps}
        step.close()
    except Exception as e:   # noqa: BLE001
        print(f"[bench] graph step on {name} failed: {e}", file=sys.stderr)
    def per_call(calls, sync):
        """every call timed on its own (device outputs: synchronised per call): (median seconds, max seconds) -- a single stalled call in
        a loop of 20-50 must not pass for the call's cost, as a bare mean made it do (round 2: graph replay; round 3: a 0.85 ms C4 call)"""
        ts = []
        for c_ in calls:
            t0 = time.perf_counter()
            o_ = c_()
            if sync:
                torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
            del o_
        ts.sort()
        return ts[len(ts) // 2], ts[-1]

    for _ in range(3):
        ugs_sampler.sample_batch(ei_t, ptr_t, m, k, mode="sample", seed=42)
    dt_host, dt_host_max = per_call([lambda: ugs_sampler.sample_batch(ei_t, ptr_t, m, k, mode="sample", seed=42)] * reps, False)
    # shuffled mini-batches: every call is a NEW combination of already-seen graphs (per-graph LRU hits, no plan to reuse): what a
    # trainer's DataLoader produces every step.  Disjoint sets of such batches, each batch timed once: (a) host-visible outputs and
    # (b) device outputs through the device batch pass, (c) / (d) the same with the pass switched off (the general host path of rounds 1-2).
    rng = np.random.default_rng(0)
    n_per = int(ptr[1] - ptr[0])
    cols_per = ei.shape[1] // G

    def make_shuffled(count):
        res = []
        for _ in range(count):
            perm = rng.permutation(G)
            blocks = [ei[:, g * cols_per:(g + 1) * cols_per] - g * n_per + i * n_per for i, g in enumerate(perm)]
            res.append(torch.from_numpy(np.ascontiguousarray(np.concatenate(blocks, axis=1))))
        return res

    def time_shuffled(batches, **kw):
        return per_call([(lambda e_s=e_s: ugs_sampler.sample_batch(e_s, ptr_t, m, k, mode="sample", seed=42, **kw)) for e_s in batches], bool(kw))[0]

    nsh = min(reps, 20)
    sets = [make_shuffled(nsh) for _ in range(3)]
    os.environ["UGS_DEVICE_BATCH"] = "1"                     # (a), (b): through the device batch pass whatever the batch's size (default: from 2048 columns on)
    try:
        time_shuffled(make_shuffled(3))                      # steady state: the graphs' root records are in the device arena, the plan cache is turning over
        time_shuffled(make_shuffled(3), device=dev)
        dt_shuf = time_shuffled(sets[0])
        dt_shuf_dev = time_shuffled(sets[1], device=dev)
        os.environ["UGS_DEVICE_BATCH"] = "0"                 # (c), (d): the general host path
        time_shuffled(make_shuffled(3))
        dt_shuf_host = time_shuffled(sets[2])
        dt_shuf_host_dev = time_shuffled(make_shuffled(nsh), device=dev)
    finally:
        os.environ.pop("UGS_DEVICE_BATCH", None)
    del sets
    # cold calls: EVERY graph of every call is new to the LRU (other dataset seeds of the same shape) -- what an epoch over a dataset
    # larger than UGS_CACHE_SIZE pays per step (QM9: always; PROTEINS at the default 1000: regularly).  Through the device pass with the
    # unknown graphs preprocessed (a) on the device (ugs_bp_roots, the default) and (b) on the host (UGS_DEVICE_COLD=0: rounds 1-3),
    # and (c) through the general host path; each batch timed once, medians.
    n_und = wl.TU_SHAPES[name][1]
    ncold = min(reps, 12)

    def fresh_batches(first_seed):
        return [torch.from_numpy(wl.tu_batch(n_per, n_und, G, dataset_seed=first_seed + t)[0]) for t in range(ncold)]

    cold = {}
    try:
        for key, env in (("device", {"UGS_DEVICE_BATCH": "1"}), ("device_pass_host_preproc", {"UGS_DEVICE_BATCH": "1", "UGS_DEVICE_COLD": "0"}),
                         ("general_path", {"UGS_DEVICE_BATCH": "0"})):
            os.environ.update(env)
            seed0 = {"device": 1000, "device_pass_host_preproc": 2000, "general_path": 3000}[key]
            time_shuffled(fresh_batches(seed0 + 500)[:3])            # the path's own buffers and code warm, on graphs of their own
            cold[key + "_ms"] = round(time_shuffled(fresh_batches(seed0)) * 1e3, 4)
            cold[key + "_device_out_ms"] = round(time_shuffled(fresh_batches(seed0 + 100), device=dev) * 1e3, 4)
            for v in env:
                os.environ.pop(v, None)
    finally:
        for v in ("UGS_DEVICE_BATCH", "UGS_DEVICE_COLD"):
            os.environ.pop(v, None)
    # device outputs: (a) calls issued back to back, synchronised once per chunk of 10 -- what a consumer on the same stream sees; the
    # median of 5 chunks, so that one stalled call does not pass for the rate -- and (b) the latency of one call with a synchronise
    chunks = []
    for _ in range(5):
        t = time.perf_counter()
        for _ in range(10):
            out_dev = ugs_sampler.sample_batch(ei_t, ptr_t, m, k, mode="sample", seed=42, device=dev)
        torch.cuda.synchronize()
        chunks.append((time.perf_counter() - t) / 10)
        del out_dev
    dt_devout = sorted(chunks)[2]
    dt_devout_lat, dt_devout_max = per_call([lambda: ugs_sampler.sample_batch(ei_t, ptr_t, m, k, mode="sample", seed=42, device=dev)] * 20, True)
    res = {"rows": rows, "k": k, "device_resident_subgraphs_per_s": round(rows / dt_dev, 1), "device_resident_ms": round(dt_dev * 1e3, 4),
           "device_resident_ms_max_rep": round(per[-1], 4), "device_resident_ms_wall_mean": round(dt_wall * 1e3, 4),
           "hip_graph_replay_subgraphs_per_s": round(rows / dt_graph, 1) if dt_graph else None,
           "hip_graph_replay_ms": round(dt_graph * 1e3, 4) if dt_graph else None, "hip_graph_replay": graph_stats,
           "drop_in_call_subgraphs_per_s": round(rows / dt_host, 1), "drop_in_call_ms": round(dt_host * 1e3, 4), "drop_in_call_ms_max": round(dt_host_max * 1e3, 4),
           "drop_in_call_device_out_latency_ms": round(dt_devout_lat * 1e3, 4), "drop_in_call_device_out_latency_ms_max": round(dt_devout_max * 1e3, 4),
           "drop_in_call_timing": "host-visible calls and the shuffled batches: median of the calls, each timed on its own (device outputs: synchronised per call); "
                                  "drop_in_call_device_out_ms: calls issued back to back, median of 5 chunks of 10",
           "drop_in_call_shuffled_batch_ms": round(dt_shuf * 1e3, 4), "drop_in_call_shuffled_batch_device_out_ms": round(dt_shuf_dev * 1e3, 4),
           "drop_in_call_shuffled_batch_general_path_ms": round(dt_shuf_host * 1e3, 4),
           "drop_in_call_shuffled_batch_general_path_device_out_ms": round(dt_shuf_host_dev * 1e3, 4), "drop_in_call_device_out_ms": round(dt_devout * 1e3, 4),
           "drop_in_call_cold": cold, "roofline": roofline}
    try:
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import oracle
        cache = oracle.Cache()
        oracle.sample_batch(ei, ptr, 1, k, "sample", 42, cache=cache)
        t = time.perf_counter()
        o = oracle.sample_batch(ei, ptr, m, k, "sample", 42, cache=cache)
        dt_cpu = time.perf_counter() - t
        g = ugs_sampler.sample_batch(ei_t, ptr_t, m, k, mode="sample", seed=42)
        res["cpu_port_subgraphs_per_s"] = round(rows / dt_cpu, 1)
        res["bit_exact_vs_cpu"] = bool(all(np.array_equal(a.numpy(), b) for a, b in zip(g, o)))
    except Exception as e:   # noqa: BLE001
        res["cpu_port_error"] = str(e)
    try:     # the reference C++ itself (oracle/_ref, prebuilt in the build container from /root/reference), if it travelled
        ref = load_prebuilt_reference()
        if ref is not None:
            ref.sample_batch(ei_t, ptr_t, 1, k, "sample", 42)            # warm its preprocessing LRU
            runs = []
            for _ in range(5):
                t = time.perf_counter()
                r = ref.sample_batch(ei_t, ptr_t, m, k, "sample", 42)
                runs.append(time.perf_counter() - t)
            runs.sort()
            best = runs[0]
            g = ugs_sampler.sample_batch(ei_t, ptr_t, m, k, mode="sample", seed=42)
            res["cpu_baseline"] = {"value": round(rows / best, 1), "unit": "k-subgraphs/s", "cores": 1, "kind": "reference",
                                   "sample": "the whole batch, best of 5, warm preprocessing LRU; oracle/_ref (the reference's own sources)",
                                   "runs_subgraphs_per_s": {"best": round(rows / runs[0], 1), "median": round(rows / runs[2], 1), "worst": round(rows / runs[-1], 1)}}
            res["reference_cpp_subgraphs_per_s"] = round(rows / best, 1)
            res["bit_exact_vs_reference_cpp"] = bool(all(torch.equal(a, b) for a, b in zip(g, r)))
    except Exception as e:   # noqa: BLE001
        res["reference_cpp_error"] = str(e)[:200]
    plan.close()
    return res


def bench_epsilon(wl, torch, reps=20):
    """epsilon_uniform_sampler.sample_batch on the PROTEINS-shaped batch (k=6, 32 x 256 samples, eps 0.1): the HIP entry
    point (host tensors in and out) next to the reference module, if its prebuilt copy travelled (oracle/_ref)."""
    import glob
    import importlib.util
    import epsilon_uniform_sampler as eps
    ei, ptr, m, k = wl.workload("c3_proteins_b8192")
    ei_t, ptr_t = torch.from_numpy(ei), torch.from_numpy(ptr)
    rows = (len(ptr) - 1) * m
    for _ in range(3):
        o = eps.sample_batch(ei_t, ptr_t, m, k, "sample", 42, 0.1)
    t = time.perf_counter()
    for r in range(reps):
        o = eps.sample_batch(ei_t, ptr_t, m, k, "sample", 42 + r, 0.1)
    dt = (time.perf_counter() - t) / reps
    res = {"rows": rows, "k": k, "epsilon": 0.1, "hip_call_ms": round(dt * 1e3, 4), "hip_subgraphs_per_s": round(rows / dt, 1),
           "failed_rows": int((o[0][:, 0] < 0).sum())}
    hits = glob.glob(os.path.join(ROOT, "oracle", "_ref", "epsilon_uniform_sampler*.so"))
    if hits:
        spec = importlib.util.spec_from_file_location("epsilon_uniform_sampler", hits[0])
        ref = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(ref)
        t = time.perf_counter()
        r = ref.sample_batch(ei_t, ptr_t, m, k, "sample", 42, 0.1)
        dt_ref = time.perf_counter() - t
        res["reference_cpp_call_ms"] = round(dt_ref * 1e3, 3)
        res["reference_cpp_subgraphs_per_s"] = round(rows / dt_ref, 1)
        res["reference_threads"] = os.cpu_count()
    return res


def load_prebuilt_reference():
    """oracle/_ref/ugs_sampler*.so if present (never built here: the reference sources do not exist on the GPU box)."""
    import glob
    import importlib.util
    hits = glob.glob(os.path.join(ROOT, "oracle", "_ref", "ugs_sampler*.so"))
    if not hits:
        return None
    import torch  # noqa: F401
    spec = importlib.util.spec_from_file_location("ugs_sampler", hits[0])
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def reference_on_headline(ei_t, ptr_t, m_total, k, mode, rows, G):
    """The reference C++ itself (oracle/_ref, built in the build container from the reference's own sources) on the headline
    workload, 1 thread.  Single graph: handle API, rows [0, rows) of the job (its preprocessing allocates a visited array per
    root, reference src/preproc.cpp:202: about half a minute at |V| = 1M, not timed).  Batch: sample_batch, warm LRU."""
    ref = load_prebuilt_reference()
    if ref is None:
        return None
    if G == 1:
        t = time.perf_counter()
        h = ref.create_preproc(ei_t, int(ptr_t[-1]), k)
        tp = time.perf_counter() - t
        t = time.perf_counter()
        ref.sample(h, rows, k, EDGE_MODE[mode], 0, 42)
        ts = time.perf_counter() - t
        ref.destroy_preproc(h)
        desc = f"rows [0,{rows}) of the same job through the reference's handle API (create_preproc {tp:.1f}s excluded), seed 42"
    else:
        ref.sample_batch(ei_t, ptr_t, 1, k, mode, 42)
        m_cpu = max(1, rows // G) if rows < m_total * G else m_total
        ts = 1e9
        for _ in range(3):
            t = time.perf_counter()
            ref.sample_batch(ei_t, ptr_t, m_cpu, k, mode, 42)
            ts = min(ts, time.perf_counter() - t)
        rows = m_cpu * G
        desc = f"the same {G}-graph batch with m_per_graph={m_cpu} ({rows} rows), best of 3, warm preprocessing LRU"
    return {"value": round(rows / ts, 1), "unit": "k-subgraphs/s", "cores": 1, "kind": "reference", "cpu_model": cpu_model(),
            "sample": desc + "; oracle/_ref = the reference's own C++ sources compiled as they are"}


def port_vs_reference_on_er_proxy(wl, torch):
    """ER proxy small enough for the reference's O(n^2) preprocessing (n = 100k, same degree law as C5): k-subgraphs/s of
    the reference C++ and of the oracle port on the same rows -- how conservative the `port` CPU baseline is."""
    ref = load_prebuilt_reference()
    if ref is None:
        return None
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle
    ei, ptr = wl.er_graph(100_000, 2_000_000, 0)
    n, k, m = 100_000, 8, 20_000
    ei_t = torch.from_numpy(ei)
    h = ref.create_preproc(ei_t, n, k)
    t = time.perf_counter()
    r = ref.sample(h, m, k, "local", 0, 42)
    t_ref = time.perf_counter() - t
    ref.destroy_preproc(h)
    P = oracle.Preproc(ei, n, k)
    t = time.perf_counter()
    o = P.sample(m, k, "local", 0, 42)
    t_port = time.perf_counter() - t
    same = bool(all(np.array_equal(a.numpy(), b) for a, b in zip(r, o)))
    return {"graph": "ER n=100k, 2M columns, k=8, 20000 rows", "reference_cpp_subgraphs_per_s": round(m / t_ref, 1),
            "port_subgraphs_per_s": round(m / t_port, 1), "port_over_reference": round(t_ref / t_port, 3), "identical_output": same,
            "cpu_baseline": {"value": round(m / t_ref, 1), "unit": "k-subgraphs/s", "cores": 1, "kind": "reference",
                             "sample": "rows [0,20000) on the ER proxy (same degree law as the headline graph), reference's handle API, preprocessing excluded"}}


if __name__ == "__main__":
    main()
