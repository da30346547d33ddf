#!/bin/bash
# kernel + memory-copy trace of the streamed host-visible call on C5 (tools/streamed_call_probe.py): the walk chunks on the job stream
# beside the device-to-host copies on the copy stream.  Output: gpurun_out/prof_stream/{kernel_stats.csv,memory_copy_stats.csv,overlap.txt}
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/prof_stream; mkdir -p $O
UGS_PROBE_ONLY_DEFAULT=1 timeout -k 10 300 rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d $O/raw -- python3 tools/streamed_call_probe.py c5_er_1m 3 > $O/probe.json 2> $O/probe.err || echo "profile failed"
for f in kernel_stats memory_copy_stats kernel_trace memory_copy_trace; do find $O/raw -name "*_$f.csv" | head -1 | xargs -I{} cp {} $O/$f.csv; done
python3 - <<'P' > $O/overlap.txt
import csv, sys
O = "gpurun_out/prof_stream"
def rows(name):
    try:
        return list(csv.DictReader(open(f"{O}/{name}.csv")))
    except Exception as e:
        print("missing", name, e); return []
k = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows("kernel_trace")]
c = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Direction", "")) for r in rows("memory_copy_trace")]
walks = sorted(x for x in k if "ugs_walk" in x[2])
d2h = sorted(x for x in c if "DEVICE_TO_HOST" in x[2].upper() and x[1] - x[0] > 200_000)   # the result copies (> 0.2 ms)
if not walks or not d2h:
    print("no walks or no copies in the trace"); sys.exit(0)
# the LAST streamed call of the run: the last 8 walk launches and the copies from the first of them on
w = walks[-8:]
t0 = w[0][0]
cc = [x for x in d2h if x[0] >= t0]
walk_busy = sum(e - s for s, e, _ in w)
copy_busy = sum(e - s for s, e, _ in cc)
both = 0
for s, e, _ in w:
    for cs, ce, _ in cc:
        both += max(0, min(e, ce) - max(s, cs))
end = max([x[1] for x in cc] + [w[-1][1]])
print(f"last streamed call: 8 walk launches {walk_busy/1e6:.3f} ms busy, {len(cc)} result copies {copy_busy/1e6:.3f} ms busy, "
      f"both at once {both/1e6:.3f} ms, first walk start -> last copy end {(end - t0)/1e6:.3f} ms")
for s, e, n in w:
    print(f"  walk  {(s - t0)/1e6:8.3f} -> {(e - t0)/1e6:8.3f} ms")
for s, e, d in cc:
    print(f"  copy  {(s - t0)/1e6:8.3f} -> {(e - t0)/1e6:8.3f} ms")
P
cat $O/overlap.txt | head -40
