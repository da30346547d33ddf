#!/usr/bin/env python3
"""bench.py -- k-subgraphs sampled per second on MI355X (BASELINE.json metric), one JSON line on rank 0.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload c5_er_1m|c2_mutag_b1024|c3_proteins_b8192|c4_qm9_b65536]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A *step* is one pass of the hot path over one batch: walk kernel(s) + scan + fill kernel, producing the reference's
tensors (nodes, edge_index, edge_ptr, edge_src) in HBM from a plan (preprocessed graph batch) that is already resident
in HBM -- the state the reference is in with a warm preprocessing LRU.  Default workload: BASELINE.json configs[4], the
Erdos-Renyi graph |V|=1M, 20M columns, k=8, 1M samples per GPU (weak scaling: N GPUs produce N*1M rows of the same job,
rank r owning rows [r*1M, (r+1)*1M); for N>1 every step also collates the batch on rank 0 over RCCL -- on a side
stream, double-buffered, so the collation of batch s overlaps the sampling of batch s+1; all K collations are inside the
timed region).
The seed changes every step (42 + step) so no step can reuse a previous step's output.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "ss-gnn_amd"))

HBM_PEAK_GBPS = 8000.0   # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md, chip table)


def log(*a):
    print(*a, file=sys.stderr, flush=True)


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


def main():
    # Libraries (RCCL prints a version banner) must not pollute stdout: fd 1 is pointed at stderr for the whole run and the
    # single JSON line goes to the saved, real stdout at the end.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    def emit(line):
        os.write(real_stdout, (line + "\n").encode())

    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="c5_er_1m")
    ap.add_argument("--mode", default="sample")
    ap.add_argument("--cpu-sample", type=int, default=-1, help="rows timed on the CPU baseline (default: sized per workload)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true")
    ap.add_argument("--force-collate", action="store_true", help="run the multi-GPU collation path even with one rank (rehearsal)")
    args = ap.parse_args()

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
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1 or args.force_collate:
        if "MASTER_ADDR" not in os.environ:
            os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", os.environ.get("MASTER_PORT", "29517")
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
        log(f"[bench] workload {args.workload}: G={G} cols={ei.shape[1]} k={k} m/GPU={m}; generate {t1 - t0:.1f}s, "
            f"preprocess+upload {t2 - t1:.1f}s, plan {info['device_bytes'] / 1e6:.0f} MB in HBM, tier {info['tier']}")
    rows_local = G * m
    m_total = m * world                      # weak scaling: every GPU adds m samples per graph to the job
    total_rows = G * m_total
    row_begin, row_count = ud.shard_range(total_rows, rank, world)
    assert row_count == rows_local

    use_collate = world > 1 or args.force_collate
    nsets = 2 if use_collate else 1          # double buffering: step s samples into set s%2 while set (s-1)%2 is collated
    nodes_bufs = [torch.empty((row_count, k), dtype=torch.int64, device=dev) for _ in range(nsets)]
    eptr_bufs = [torch.empty((row_count + 1,), dtype=torch.int64, device=dev) for _ in range(nsets)]
    nodes_buf, eptr_buf = nodes_bufs[0], eptr_bufs[0]
    # edge capacity from one synchronous probe step (+5%); identical on every rank
    _, _, tot = plan.walk(m_total, args.mode, 41, row_begin, row_count, out=(nodes_buf, eptr_buf), sync=True)
    cap_t = torch.tensor([int(tot * 1.05) + 4096], dtype=torch.int64, device=dev)
    if world > 1:
        dist.all_reduce(cap_t, op=dist.ReduceOp.MAX)
    cap = int(cap_t.item())
    eidx_bufs = [torch.empty((2, cap), dtype=torch.int64, device=dev) for _ in range(nsets)]
    esrc_bufs = [torch.empty((cap,), dtype=torch.int64, device=dev) for _ in range(nsets)]
    totals = torch.zeros((args.steps + args.warmup + 1,), dtype=torch.int64, device=dev)
    node_bound = int(ptr[-1])
    edge_bound = max(node_bound, m_total * k)
    main_stream = torch.cuda.current_stream()
    side = torch.cuda.Stream(device=dev) if use_collate else None
    ev_sampled = [torch.cuda.Event() for _ in range(nsets)]
    ev_collated = [torch.cuda.Event() for _ in range(nsets)]
    collated_once = [False] * nsets

    def sample(i):
        """walk + scan + fill of step i into buffer set i % nsets (asynchronous, main stream)"""
        b = i % nsets
        if use_collate and collated_once[b]:
            main_stream.wait_event(ev_collated[b])            # the set is free once its previous batch has been collated
        plan.walk(m_total, args.mode, 42 + i, row_begin, row_count, out=(nodes_bufs[b], eptr_bufs[b]), sync=False)
        plan.fill(m_total, nodes_bufs[b], eptr_bufs[b], None, args.mode, row_begin, out=(eidx_bufs[b], esrc_bufs[b]))
        totals[i] = eptr_bufs[b][-1]
        if use_collate:
            ev_sampled[b].record(main_stream)

    def collate_step(i):
        """the one exchange step: collate batch i on rank 0 (side stream, overlaps the sampling of batch i+1)"""
        b = i % nsets
        with torch.cuda.stream(side):
            side.wait_event(ev_sampled[b])
            res = ud.collate((nodes_bufs[b], eidx_bufs[b], eptr_bufs[b], esrc_bufs[b]), k, args.mode, node_bound, edge_bound,
                             ei.shape[1], dst=0)
            ev_collated[b].record(side)
        collated_once[b] = True
        return res

    def run_steps(first, count):
        """`count` complete steps: every batch sampled AND (multi-GPU) collated inside the call"""
        if count <= 0:
            return
        for i in range(first, first + count):
            sample(i)
            if use_collate and i > first:
                collate_step(i - 1)
        if use_collate:
            collate_step(first + count - 1)
            side.synchronize()

    run_steps(0, args.warmup)
    torch.cuda.synchronize()
    plan.set_timing(True)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t_start = time.perf_counter()
    run_steps(args.warmup, args.steps)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t_start
    el_t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(el_t, op=dist.ReduceOp.MAX)
    elapsed = float(el_t.item())
    timing = plan.get_timing()
    plan.set_timing(False)
    tmax = int(totals.max().item())
    if tmax > cap:
        raise SystemExit(f"edge capacity {cap} too small for {tmax}: result invalid")

    if rank != 0:
        dist.barrier()
        dist.destroy_process_group()
        return

    value = total_rows * args.steps / elapsed
    ms_per_step = elapsed / args.steps * 1e3
    walk_ms = timing["walk"][0] / max(timing["walk"][1], 1)
    scan_ms = timing["scan"][0] / max(timing["scan"][1], 1)
    fill_ms = timing["fill"][0] / max(timing["fill"][1], 1)
    launch = plan.last_launch()

    # ---- algorithmic bytes from a reference step (seed 42) and parity of its first rows against the CPU oracle -----
    plan.walk(m_total, args.mode, 42, row_begin, row_count, out=(nodes_buf, eptr_buf), sync=True)
    nodes_h = nodes_buf.cpu().numpy()
    eptr_h = eptr_buf.cpu().numpy()
    deg = csr_degrees(ei, node_bound)
    walk_bytes, fill_bytes = split_algorithmic_bytes(nodes_h, eptr_h, k, deg)
    unit_bytes = walk_bytes + fill_bytes
    achieved = walk_bytes * row_count / (walk_ms * 1e-3) / 1e9 if walk_ms > 0 else 0.0
    gpu_ms = walk_ms + scan_ms + fill_ms
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")   # written from separate rocprofv3 --pmc passes of this command
    if os.path.exists(tpath):
        try:
            with open(tpath) as f:
                traffic = json.load(f).get(args.workload, {}).get("walk_kernel_hbm_bytes_per_launch")
        except Exception:
            traffic = None
    roofline = {"bound": "hbm", "kernel": launch["kernel"], "achieved": round(achieved, 2), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBPS, 5), "traffic": traffic,
                "algorithmic_bytes_per_unit": round(walk_bytes, 1), "units_per_launch": row_count,
                "kernel_ms": round(walk_ms, 4), "grid": launch["grid"], "block": launch["block"], "lds_bytes_per_block": launch["lds_bytes"],
                "path": {"algorithmic_bytes_per_unit": round(unit_bytes, 1), "gpu_ms_per_step": round(gpu_ms, 4),
                         "achieved": round(unit_bytes * row_count / (gpu_ms * 1e-3) / 1e9, 2) if gpu_ms > 0 else 0.0,
                         "fill_kernel_ms": round(fill_ms, 4), "scan_ms": round(scan_ms, 4)}}

    cpu_baseline = None
    parity_rows = 0
    if not args.no_cpu_baseline and world == 1:
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import oracle   # the checker, used here only as the reported CPU baseline and for a parity spot check
        n_cpu = args.cpu_sample if args.cpu_sample > 0 else (100_000 if args.workload.startswith("c5") else min(total_rows, 400_000))
        if G == 1:
            tp = time.time()
            P = oracle.Preproc(ei, int(ptr[1]), k)
            tp = time.time() - tp
            n_cpu = min(n_cpu, m_total)
            tc = time.perf_counter()
            o_nodes, o_eidx, o_eptr, o_esrc = P.sample(m_total, k, {"sample": "local", "graph": "flat", "global": "global"}[args.mode], 0, 42, 0, n_cpu)
            tc = time.perf_counter() - tc
            sample_desc = f"rows [0,{n_cpu}) of the same {m_total}-row job on the same graph, seed 42; preprocessing ({tp:.1f}s) excluded like the warm-cache GPU plan"
            parity_rows = n_cpu
            assert np.array_equal(o_nodes, nodes_h[:n_cpu]) and np.array_equal(o_eptr, eptr_h[: n_cpu + 1]), "GPU rows differ from the CPU oracle"
        else:
            cache = oracle.Cache()
            oracle.sample_batch(ei, ptr, 1, k, args.mode, 42, cache=cache)          # warm the LRU (preprocessing excluded)
            m_cpu = max(1, min(m_total, n_cpu // G))
            tc = time.perf_counter()
            o = oracle.sample_batch(ei, ptr, m_cpu, k, args.mode, 42, cache=cache)
            tc = time.perf_counter() - tc
            n_cpu = m_cpu * G
            sample_desc = f"the same {G}-graph batch with m_per_graph={m_cpu} ({n_cpu} rows), seed 42, warm preprocessing LRU"
            if m_cpu == m_total:
                parity_rows = n_cpu
                assert np.array_equal(o[0], nodes_h) and np.array_equal(o[2], eptr_h), "GPU rows differ from the CPU oracle"
        cpu_baseline = {"value": round(n_cpu / tc, 1), "unit": "k-subgraphs/s", "cores": 1, "kind": "port",
                        "sample": sample_desc + "; oracle/ugs_oracle.c (plain-C restatement of the reference algorithm), 1 thread"}

    out = {"metric": "k_subgraphs_sampled_per_sec", "value": round(value, 1), "unit": "k-subgraphs/s", "n_gpus": world,
           "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
           "scaling": "weak", "vs_baseline": None, "dtype": "int64", "data": "synthetic",
           "config": {"workload": args.workload, "graphs": G, "columns": int(ei.shape[1]), "k": k, "rows_per_gpu": rows_local,
                      "global_rows": total_rows, "mode": args.mode, "sharding": f"rows{world}" if world > 1 else "none",
                      "collate": "gather to rank 0 over RCCL every step, overlapped with the next step's sampling" if use_collate else "none (single GPU)"},
           "roofline": roofline, "cpu_baseline": cpu_baseline, "parity_checked_rows": parity_rows}

    # ---- secondary workloads (single GPU only, quick): the TU-shaped configurations of BASELINE.json -----------------
    if not args.no_extras and world == 1:
        extras = {}
        for name in ("c2_mutag_b1024", "c3_proteins_b8192", "c4_qm9_b65536"):
            if name == args.workload:
                continue
            try:
                extras[name] = bench_small(name, ugs_sampler, wl, torch, dev)
            except Exception as e:   # noqa: BLE001
                extras[name] = {"error": str(e)}
        out["other_workloads"] = extras
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
    sys.stdout.flush()
    emit(json.dumps(out))
    if world > 1 or args.force_collate:
        dist.barrier()
        dist.destroy_process_group()


def bench_small(name, ugs_sampler, wl, torch, dev, reps=50):
    """TU-shaped configuration: (a) device-resident plan path, outputs in HBM; (b) the drop-in host call
    ugs_sampler.sample_batch(...) end to end (slice + hash + LRU lookups, kernels, D2H into pinned tensors)."""
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
        plan.walk(m, "sample", 42 + i, 0, rows, out=(nodes, eptr), sync=False)
        plan.fill(m, nodes, eptr, None, "sample", 0, out=(eidx, esrc))
    torch.cuda.synchronize()
    t = time.perf_counter()
    for i in range(reps):
        plan.walk(m, "sample", 42 + i, 0, rows, out=(nodes, eptr), sync=False)
        plan.fill(m, nodes, eptr, None, "sample", 0, out=(eidx, esrc))
    torch.cuda.synchronize()
    dt_dev = (time.perf_counter() - t) / reps
    # the same step captured once as a HIP graph and replayed (Plan.graph_step)
    dt_graph = None
    try:
        step = plan.graph_step(m, "sample", 0, rows, edge_capacity=cap)
        for i in range(5):
            step.launch(42 + i)
        torch.cuda.synchronize()
        t = time.perf_counter()
        for i in range(reps):
            step.launch(42 + i)
        torch.cuda.synchronize()
        dt_graph = (time.perf_counter() - t) / reps
        step.close()
    except Exception as e:   # noqa: BLE001
        print(f"[bench] graph step on {name} failed: {e}", file=sys.stderr)
    for _ in range(3):
        ugs_sampler.sample_batch(ei_t, ptr_t, m, k, mode="sample", seed=42)
    t = time.perf_counter()
    for _ in range(reps):
        ugs_sampler.sample_batch(ei_t, ptr_t, m, k, mode="sample", seed=42)
    dt_host = (time.perf_counter() - t) / reps
    # shuffled mini-batches: every call is a NEW combination of already-seen graphs (per-graph LRU hits, plan-cache miss)
    rng = np.random.default_rng(0)
    n_per = int(ptr[1] - ptr[0])
    cols_per = ei.shape[1] // G
    shuffled = []
    for _ in range(min(reps, 20)):
        perm = rng.permutation(G)
        blocks = [ei[:, g * cols_per:(g + 1) * cols_per] - g * n_per + i * n_per for i, g in enumerate(perm)]
        shuffled.append(torch.from_numpy(np.ascontiguousarray(np.concatenate(blocks, axis=1))))
    t = time.perf_counter()
    for e_s in shuffled:
        ugs_sampler.sample_batch(e_s, ptr_t, m, k, mode="sample", seed=42)
    dt_shuf = (time.perf_counter() - t) / len(shuffled)
    t = time.perf_counter()
    for _ in range(reps):
        out_dev = ugs_sampler.sample_batch(ei_t, ptr_t, m, k, mode="sample", seed=42, device=dev)
    torch.cuda.synchronize()
    dt_devout = (time.perf_counter() - t) / reps
    res = {"rows": rows, "k": k, "device_resident_subgraphs_per_s": round(rows / dt_dev, 1), "device_resident_ms": round(dt_dev * 1e3, 4),
           "hip_graph_replay_subgraphs_per_s": round(rows / dt_graph, 1) if dt_graph else None,
           "hip_graph_replay_ms": round(dt_graph * 1e3, 4) if dt_graph else None,
           "drop_in_call_subgraphs_per_s": round(rows / dt_host, 1), "drop_in_call_ms": round(dt_host * 1e3, 4),
           "drop_in_call_shuffled_batch_ms": round(dt_shuf * 1e3, 4), "drop_in_call_device_out_ms": round(dt_devout * 1e3, 4)}
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
            best = 1e9
            for _ in range(3):
                t = time.perf_counter()
                r = ref.sample_batch(ei_t, ptr_t, m, k, "sample", 42)
                best = min(best, time.perf_counter() - t)
            g = ugs_sampler.sample_batch(ei_t, ptr_t, m, k, mode="sample", seed=42)
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
            "port_subgraphs_per_s": round(m / t_port, 1), "port_over_reference": round(t_ref / t_port, 3), "identical_output": same}


if __name__ == "__main__":
    main()
