#!/usr/bin/env python3
"""HBM traffic of the walk kernel per launch from the rocprofv3 --pmc passes of tools/profile_pmc.sh (FETCH_SIZE and WRITE_SIZE in
their own runs) -> profiles/pmc_traffic.json, which bench.py reports as roofline.traffic.

gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE counts every read request as 64 bytes.  Calibrated for this kernel's
access shapes with tools/calib_fetch.hip (profiles/r02_calib_fetch_counter_collection.csv):
  * wide contiguous reads (8 or 16 B per lane streams; adjacency rows spanning whole 128-byte lines) leave L2 as 128-byte requests
    and are counted at exactly 1/2 -> x2;
  * a single 8-byte gather from a random line is served by ONE 64-byte request (k_gather: 63.9 B counted per gather; touching both
    halves of the line costs a second request) -> counted exactly, x1.
The walk kernel's single gathers are known in number from its own output: one per staged hit (the edge column of every undirected
induced edge = edge entries / 2) and the 24-byte root record (1.25 sectors on average: records at 24-byte stride, 2 of 8 alignments
straddle a 64-byte boundary).  bytes = 2 * (FETCH - gathers * 64) + gathers * 64 + WRITE.
usage: tools/traffic_from_pmc.py <pmc summary json> <bench json of the same build> [workload]"""
import json, sys
summ, bench = json.load(open(sys.argv[1])), json.load(open(sys.argv[2]))
wl = sys.argv[3] if len(sys.argv) > 3 else bench["config"]["workload"]
k = [x for x in summ if "ugs_walk_lds<64, 448" in x][0]
fetch = summ[k]["FETCH_SIZE"]["mean"] * 1024
write = summ[k]["WRITE_SIZE"]["mean"] * 1024
rows = bench["roofline"]["units_per_launch"]
alg = bench["roofline"]["algorithmic_bytes_per_unit"]
es = (bench["roofline"]["path"]["algorithmic_bytes_per_unit"] - alg) / 28.0          # edge entries per row
gathers = rows * (es / 2.0 + 1.25)
single = gathers * 64
total = 2 * (fetch - single) + single + write
out = {"_note": __doc__.split("usage:")[0].strip(),
       wl: {"kernel": bench["roofline"]["kernel"], "FETCH_SIZE_KB": summ[k]["FETCH_SIZE"]["mean"], "WRITE_SIZE_KB": summ[k]["WRITE_SIZE"]["mean"],
            "single_64B_gathers_per_launch": round(gathers), "edge_entries_per_row": round(es, 2),
            "walk_kernel_hbm_bytes_per_launch": round(total), "units_per_launch": rows, "algorithmic_bytes_per_launch": round(alg * rows),
            "ratio_to_algorithmic": round(total / (alg * rows), 3),
            "uncalibrated_2xFETCH_plus_WRITE": round(2 * fetch + write),
            "why_above_algorithmic": "line granularity, not re-reads: the header + first lines of a padded row are fetched whole (3 of the block's 4 lines "
                                     "with the header, the 4th only for rows that reach into it), the root record costs a 64-byte sector or two, every staged hit's edge "
                                     "column one sector, and the walk writes ~14 staged 8-byte items per row beside nodes and counts",
            "source": sys.argv[1]}}
# what binds the kernel is instruction issue, not HBM: the same passes give the instruction mix per walk and how busy the issue ports are
c = summ[k]
def cnt(name):
    return c[name]["mean"] if name in c else None
if cnt("SQ_INSTS_VALU") is not None and cnt("SQ_BUSY_CYCLES") is not None:
    simd_quads = cnt("SQ_BUSY_CYCLES") / 32.0 / 4.0 * 1024.0          # SQ_BUSY_CYCLES is summed over 32 shader engines; 1024 SIMDs; quad-cycles
    issue = {"instructions_per_walk": round(cnt("SQ_INSTS") / rows, 1) if cnt("SQ_INSTS") else None,
             "valu_per_walk": round(cnt("SQ_INSTS_VALU") / rows, 1), "salu_per_walk": round(cnt("SQ_INSTS_SALU") / rows, 1),
             "lds_per_walk": round(cnt("SQ_INSTS_LDS") / rows, 1) if cnt("SQ_INSTS_LDS") else None,
             "branch_per_walk": round(cnt("SQ_INSTS_BRANCH") / rows, 1) if cnt("SQ_INSTS_BRANCH") else None,
             "valu_busy_share_of_simd_quad_cycles": round(cnt("SQ_ACTIVE_INST_VALU") / simd_quads, 3) if cnt("SQ_ACTIVE_INST_VALU") else None,
             "scalar_busy_share_of_simd_quad_cycles": round(cnt("SQ_ACTIVE_INST_SCA") / simd_quads, 3) if cnt("SQ_ACTIVE_INST_SCA") else None,
             "wave_cycles_waiting": round(cnt("SQ_WAIT_ANY") / cnt("SQ_WAVE_CYCLES"), 3) if cnt("SQ_WAIT_ANY") and cnt("SQ_WAVE_CYCLES") else None,
             "wave_cycles_issue_stalled": round(cnt("SQ_WAIT_INST_ANY") / cnt("SQ_WAVE_CYCLES"), 3) if cnt("SQ_WAIT_INST_ANY") and cnt("SQ_WAVE_CYCLES") else None,
             "marginal_cost_wave_cycles": {"scalar": 12.4, "vector_4B": 7.2, "vector_8B": 9.9, "source": "profiles/r04_pad_probe.txt: 448 dummy instructions per walk of each kind"}}
    out[wl]["issue"] = issue
try:                                   # keep the other workloads' entries
    old = json.load(open("profiles/pmc_traffic.json"))
    old.update(out)
    out = old
except Exception:
    pass
json.dump(out, open("profiles/pmc_traffic.json", "w"), indent=1)
print(json.dumps(out[wl], indent=1))
