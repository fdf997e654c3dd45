#!/usr/bin/env python3
"""Static instruction counts of one kernel per SOURCE FUNCTION (line tables): compile pt_kernels.hip with -gline-tables-only,
attribute every instruction of the kernel to the source line its .loc directive names, and add the lines up per function of
pt_kernels.hip / pt_arith.inc.  Classes as priced by profiles/r03_ubench_valu.txt: A full-rate VALU (f32 add / mul / fma, add_u32,
xor, mov: 2.3 SIMD cycles), T transcendental (8.2), B every other VALU (4.2).
usage: tools/isa_regions.py [kernel-substring] [arith 0|1|2]"""
import collections, os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "cosc_4397_pathtracing_raytracing_project_amd", "csrc")
want = sys.argv[1] if len(sys.argv) > 1 else "k_bounceILb1EE"
arith = sys.argv[2] if len(sys.argv) > 2 else "2"
out = os.path.join(ROOT, "build", "scratch", f"pt_regions_{arith}.s")
os.makedirs(os.path.dirname(out), exist_ok=True)
flags = ["-ffp-contract=off"] if arith == "0" else ["-ffp-contract=fast-honor-pragmas"]
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-slp-vectorize", *flags, f"-DPT_ARITH={arith}",
                       "-gline-tables-only", "--cuda-device-only", "-S", "-o", out, os.path.join(SRC, "pt_kernels.hip")], stderr=subprocess.DEVNULL)
s = open(out).read()
m = re.search(r"^(_ZN3ptk\S*" + re.escape(want) + r"\S*):(.*?)s_endpgm", s, re.S | re.M)
body = m.group(2)
files = {int(f.group(1)): (f.group(3) or f.group(2)).split("/")[-1] for f in re.finditer(r'\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', s)}
# function start lines of the two sources
funcs = {}
for fn in ("pt_kernels.hip", "pt_arith.inc", "pt_ieee.inc", "pt_shade.inc", "pt_grid.inc", "pt_output.inc", "pt_launch.inc", "pt_ieee_check.inc"):
    starts = []
    for i, l in enumerate(open(os.path.join(SRC, fn)).read().splitlines(), 1):
        mm = re.match(r"^(?:template.*?>\s*)?(?:PT_DEV|__global__|__host__ __device__ inline|static PT_DEV|inline)\b.*?\b(\w+)\s*\(", l)
        if mm and not l.startswith(" "):
            starts.append((i, mm.group(1)))
    funcs[fn] = starts
def func_of(fn, line):
    name = f"{fn}:?"
    for i, n in funcs.get(fn, []):
        if i <= line:
            name = n
        else:
            break
    return name
A = {"v_fma_f32", "v_fmac_f32_e32", "v_mul_f32_e32", "v_add_f32_e32", "v_sub_f32_e32", "v_subrev_f32_e32", "v_add_u32_e32", "v_xor_b32_e32", "v_mov_b32_e32",
     "v_mul_f32_e64", "v_add_f32_e64", "v_sub_f32_e64", "v_fmac_f32_e64", "v_fma_f32_e64", "v_sub_u32_e32", "v_subrev_u32_e32", "v_and_b32_e32", "v_or_b32_e32"}
T = {"v_rcp_f32_e32", "v_rsq_f32_e32", "v_sqrt_f32_e32", "v_sin_f32_e32", "v_cos_f32_e32", "v_rcp_iflag_f32_e32"}
agg = collections.defaultdict(lambda: [0, 0, 0, 0, 0, 0])
loc = ("?", 0)
for l in body.splitlines():
    t = l.strip()
    mm = re.match(r"\.loc\s+(\d+)\s+(\d+)", t)
    if mm:
        loc = (files.get(int(mm.group(1)), "?"), int(mm.group(2)))
        continue
    if not t or t[0] in ";." or re.match(r"^\.?LBB", t):
        continue
    op = t.split()[0]
    a = agg[func_of(*loc) if loc[0] in funcs else loc[0]]
    if op in A: a[0] += 1
    elif op in T: a[2] += 1
    elif op.startswith("v_"): a[1] += 1
    elif op.startswith("s_"): a[3] += 1
    elif op.startswith("ds_"): a[4] += 1
    else: a[5] += 1
print(f"{'function':34s}    A    B    T SALU   DS VMEM  valu-cycles   (static, whole kernel {want}, arith {arith})")
for k, v in sorted(agg.items(), key=lambda kv: -(kv[1][0] * 2.3 + kv[1][1] * 4.2 + kv[1][2] * 8.2)):
    print(f"{k:34s} {v[0]:4d} {v[1]:4d} {v[2]:4d} {v[3]:4d} {v[4]:4d} {v[5]:4d}  {v[0]*2.3+v[1]*4.2+v[2]*8.2:9.0f}")
tot = [sum(v[i] for v in agg.values()) for i in range(6)]
print(f"{'TOTAL':34s} {tot[0]:4d} {tot[1]:4d} {tot[2]:4d} {tot[3]:4d} {tot[4]:4d} {tot[5]:4d}")
