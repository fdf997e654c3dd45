#!/usr/bin/env python3
"""Compile pt_kernels.hip to gfx950 assembly and print instruction counts per basic block of one kernel.
usage: tools/isa_blocks.py [kernel-substring] [min-instrs]"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "cosc_4397_pathtracing_raytracing_project_amd", "csrc", "pt_kernels.hip")
out = os.path.join(ROOT, "build", "scratch", "pt_kernels.s")
os.makedirs(os.path.dirname(out), exist_ok=True)
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-slp-vectorize",
                       "--cuda-device-only", "-S", "-o", out, src], stderr=subprocess.DEVNULL)
want = sys.argv[1] if len(sys.argv) > 1 else "k_intersectILb1"
minn = int(sys.argv[2]) if len(sys.argv) > 2 else 12
lines = open(out).read().split("\n")
inside = False
blocks = []
cur = None
for l in lines:
    if re.match(r"^_Z\w+:", l):
        inside = want in l
        cur = ["entry", "", 0, 0, 0, 0]
        if inside:
            blocks.append(cur)
        continue
    if not inside:
        continue
    if "s_endpgm" in l:
        inside = False
        continue
    m = re.match(r"^(\.LBB\d+_\d+):(.*)", l)
    if m:
        cur = [m.group(1), m.group(2).strip(), 0, 0, 0, 0]
        blocks.append(cur)
        continue
    t = l.strip()
    if not t or t[0] in ";.":
        continue
    op = t.split()[0]
    if op.startswith("v_"):
        cur[2] += 1
    elif op.startswith("s_"):
        cur[3] += 1
    elif op.startswith("ds_"):
        cur[4] += 1
    elif op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        cur[5] += 1
tot = [sum(b[i] for b in blocks) for i in (2, 3, 4, 5)]
print("total VALU %d SALU %d LDS %d VMEM %d" % tuple(tot))
for b in blocks:
    if sum(b[2:]) >= minn:
        print(f"{b[0]:10s} VALU {b[2]:4d} SALU {b[3]:3d} LDS {b[4]:3d} VMEM {b[5]:2d} | {b[1][:80]}")
