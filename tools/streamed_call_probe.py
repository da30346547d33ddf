"""The host-visible sample_batch call on a large workload, two-phase against streamed (ugs_sample_batch_stream), chunk sizes swept.
usage: python tools/streamed_call_probe.py [workload] [reps]   -> one JSON line; the streamed tensors are compared with the two-phase ones"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ss-gnn_amd"))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import ugs_sampler  # noqa: E402
import ugs_workloads as wl  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "c5_er_1m"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
ei, ptr, m, k = wl.workload(name)
ei_t, ptr_t = torch.from_numpy(ei), torch.from_numpy(ptr)
rows = (len(ptr) - 1) * m


def timed(seed0=42):
    ts = []
    for r in range(reps):
        t = time.perf_counter()
        o = ugs_sampler.sample_batch(ei_t, ptr_t, m, k, seed=seed0 + r)
        ts.append((time.perf_counter() - t) * 1e3)
        del o
    ts.sort()
    return {"median_ms": round(ts[len(ts) // 2], 3), "min_ms": round(ts[0], 3), "max_ms": round(ts[-1], 3)}


out = {"workload": name, "rows": rows, "reps": reps}
os.environ["UGS_NO_STREAMED_CALL"] = "1"
want = ugs_sampler.sample_batch(ei_t, ptr_t, m, k, seed=42)
out["two_phase"] = timed()
del os.environ["UGS_NO_STREAMED_CALL"]
ugs_sampler.sample_batch(ei_t, ptr_t, m, k, seed=41)
got = ugs_sampler.sample_batch(ei_t, ptr_t, m, k, seed=42)
out["streamed_equals_two_phase"] = all(torch.equal(a, b) for a, b in zip(got, want))
out["total_edges"] = int(want[1].shape[1])
del got, want
out["streamed"] = {}
for chunk in ((0,) if os.environ.get("UGS_PROBE_ONLY_DEFAULT") else (0, 31250, 62500, 250000, 500000)):
    if chunk:
        os.environ["UGS_STREAM_CHUNK_ROWS"] = str(chunk)
    ugs_sampler.sample_batch(ei_t, ptr_t, m, k, seed=41)
    out["streamed"]["default(rows/8)" if not chunk else str(chunk)] = timed()
os.environ.pop("UGS_STREAM_CHUNK_ROWS", None)
if len(ptr) == 2 and not os.environ.get("UGS_PROBE_ONLY_DEFAULT"):      # one graph: the same job through the handle API (create_preproc + sample)
    h = ugs_sampler.create_preproc(ei_t, int(ptr[-1]), k)

    def timed_handle():
        ts = []
        for r in range(reps):
            t = time.perf_counter()
            o = ugs_sampler.sample(h, m, k, "local", 0, 42 + r)
            ts.append((time.perf_counter() - t) * 1e3)
            del o
        ts.sort()
        return {"median_ms": round(ts[len(ts) // 2], 3), "min_ms": round(ts[0], 3), "max_ms": round(ts[-1], 3), "subgraphs_per_s": round(rows / ts[len(ts) // 2] * 1e3, 1)}
    os.environ["UGS_NO_STREAMED_CALL"] = "1"
    want = ugs_sampler.sample(h, m, k, "local", 0, 42)
    out["handle_two_phase"] = timed_handle()
    del os.environ["UGS_NO_STREAMED_CALL"]
    ugs_sampler.sample(h, m, k, "local", 0, 41)
    got = ugs_sampler.sample(h, m, k, "local", 0, 42)
    out["handle_streamed_equals_two_phase"] = all(torch.equal(a, b) for a, b in zip(got, want))
    del got, want
    out["handle_streamed"] = timed_handle()
    ugs_sampler.destroy_preproc(h)
for d in out["streamed"].values():
    d["subgraphs_per_s"] = round(rows / d["median_ms"] * 1e3, 1)
out["two_phase"]["subgraphs_per_s"] = round(rows / out["two_phase"]["median_ms"] * 1e3, 1)
print(json.dumps(out))
