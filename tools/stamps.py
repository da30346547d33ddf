import os, sys, ctypes, time
"""Per-phase wave-cycle totals of the walk kernel from the diagnostic build (tools/mkvar.sh stamps -DUGS_STAMPS -> ab/lib_stamps.so).
usage: python tools/stamps.py [workload]"""
os.environ["UGS_MI355_LIB"] = os.path.abspath(os.environ.get("STAMPS_LIB", "ab/lib_stamps.so"))
sys.path.insert(0, "ss-gnn_amd")
import torch, numpy as np, ugs_sampler, ugs_workloads as wl
from ugs_sampler._lib import lib
name = sys.argv[1] if len(sys.argv) > 1 else "c5_er_1m"
ei, ptr, m, k = wl.workload(name)
G = len(ptr) - 1
plan = ugs_sampler.Plan.from_batch(torch.from_numpy(ei), torch.from_numpy(ptr), k)
rows = G * m
buf = (ctypes.c_ulonglong * 32)()
lib.ugs_debug_read_stamps.argtypes = [ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int]
plan.walk(m, "sample", 42, 0, rows, sync=True)
lib.ugs_debug_read_stamps(buf, 1)
torch.cuda.synchronize(); t = time.time()
plan.walk(m, "sample", 43, 0, rows, sync=True)
dt = time.time() - t
lib.ugs_debug_read_stamps(buf, 1)
M = 1 << 64
val = lambda i: buf[i] if buf[i] < M // 2 else buf[i] - M
names = {0: "root + hash reset", 1: "scan_row (adjacency, hash, append)", 2: "select: dispatch + row-pointer issue", 3: "mark pick in hash + sample list", 4: "draw (u64 mod)",
         5: "row output + edge staging", 6: "scan_row: hash probe loop", 7: "find position of the pick", 8: "shift candidate list", 10: "select: materialise, 1 element/lane", 11: "select: materialise, 2 elements/lane", 12: "select: materialise, LDS table",
         13: "select: final, 1 element/lane", 14: "select: final, 2 elements/lane", 15: "select: final, LDS table"}
tot = sum(val(i) for i in names)
n = rows
print(f"{name} [{os.path.basename(os.environ['UGS_MI355_LIB'])}]: rows={n} launch wall={dt*1e3:.2f} ms  wave cycles/walk={tot/n:.0f}  {plan.last_launch()}")
for i, nm in names.items():
    print(f"  {i:2d} {nm:42s} {val(i)/n:9.0f} cycles/walk {100*val(i)/max(tot,1):5.1f}%   {buf[16+i]/n:6.2f} executions/walk  {val(i)/max(buf[16+i],1):8.0f} cycles each")
