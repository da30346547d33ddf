#!/bin/bash
# LDS-array counters of the walk kernel in one rocprofv3 --pmc pass: usage tools/profile_lds.sh <tag> [bench args...]
# SQ_LDS_IDX_ACTIVE (cycles the LDS is in use) and SQ_LDS_BANK_CONFLICT (cycles it is stalled by conflicts) are both in CYCLES;
# SQ_ACTIVE_INST_LDS is per-WAVE time in quad-cycles (not pipe utilisation).  GRBM_GUI_ACTIVE is summed over the 8 XCDs.
set -o pipefail
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/lds_$TAG; mkdir -p $OUT
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/p -- python3 bench.py --steps 2 --warmup 1 --no-extras --no-cpu-baseline "$@" > $OUT/bench.json 2> $OUT/bench.err
python3 - "$OUT" <<'PY'
import csv, glob, json, sys
out = sys.argv[1]
agg = {}
for path in glob.glob(out + "/p/*/*counter_collection.csv"):
    for r in csv.DictReader(open(path)):
        if "ugs_walk_lds" not in r["Kernel_Name"]:
            continue
        k = r["Kernel_Name"].split("::")[-1].split("(")[0]
        agg.setdefault(k, {}).setdefault(r["Counter_Name"], []).append((float(r["Counter_Value"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
res = {}
for k, cs in agg.items():
    m = {c: sum(v for v, _ in vals) / len(vals) for c, vals in cs.items()}
    ns = sum(t for _, t in cs["SQ_INSTS_LDS"]) / len(cs["SQ_INSTS_LDS"])
    if ns < 100000:
        continue
    cu_cycles = m["GRBM_GUI_ACTIVE"] / 8.0 * 256          # elapsed shader cycles x CUs
    res[k] = dict(m, launch_ns=ns, launches=len(cs["SQ_INSTS_LDS"]),
                  lds_busy_share_of_cu_cycles=round(m["SQ_LDS_IDX_ACTIVE"] / cu_cycles, 4),
                  bank_conflict_share_of_cu_cycles=round(m["SQ_LDS_BANK_CONFLICT"] / cu_cycles, 4),
                  bank_conflict_share_of_lds_busy=round(m["SQ_LDS_BANK_CONFLICT"] / m["SQ_LDS_IDX_ACTIVE"], 4),
                  lds_cycles_per_instruction=round(m["SQ_LDS_IDX_ACTIVE"] / m["SQ_INSTS_LDS"], 2))
print(json.dumps(res, indent=1))
json.dump(res, open(out + "/summary.json", "w"), indent=1)
PY
