#!/usr/bin/env python3
"""Static instruction mix of the probe kernels of tools/isa_probe.hip (straight-line routines: static ~ executed).
usage: tools/isa_probe.py  -> table of VALU / SALU / LDS / VMEM / branch counts per routine, minus the empty probe."""
import os, re, subprocess, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = "/tmp/probe/isa_probe.s"
os.makedirs("/tmp/probe", exist_ok=True)
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-x", "hip", "-S", "--cuda-device-only",
                os.path.join(ROOT, "tools", "isa_probe.hip"), "-o", out] + sys.argv[1:], check=True, stderr=subprocess.DEVNULL)
cur, stats = None, collections.OrderedDict()
for line in open(out):
    m = re.match(r"^(_ZN\S*probe_\S+):", line)
    if m:
        cur = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
        cur = re.sub(r"^(void )?\(anonymous namespace\)::", "", cur).split("(")[0]
        stats[cur] = collections.Counter()
        continue
    if cur is None:
        continue
    t = line.strip()
    if t.startswith("s_endpgm"):
        cur = None
        continue
    op = t.split()[0] if t and not t.startswith((";", ".")) else ""
    if not op or op.endswith(":"):
        continue
    c = stats[cur]
    if op.startswith("v_"): c["valu"] += 1
    elif op.startswith(("s_cbranch", "s_branch")): c["branch"] += 1
    elif op.startswith("s_waitcnt"): c["wait"] += 1
    elif op.startswith("s_nop"): c["nop"] += 1
    elif op.startswith("s_"): c["salu"] += 1
    elif op.startswith("ds_"): c["lds"] += 1
    elif op.startswith(("global_", "flat_", "buffer_", "scratch_")): c["vmem"] += 1
base = stats.get("probe_empty", collections.Counter())
print(f"{'routine':28s} {'VALU':>6s} {'SALU':>6s} {'LDS':>5s} {'VMEM':>5s} {'br':>4s} {'wait':>5s} {'nop':>5s}")
for k, c in stats.items():
    d = {x: c[x] - base[x] for x in ("valu", "salu", "lds", "vmem", "branch", "wait", "nop")}
    print(f"{k:28s} {d['valu']:6d} {d['salu']:6d} {d['lds']:5d} {d['vmem']:5d} {d['branch']:4d} {d['wait']:5d} {d['nop']:5d}")
