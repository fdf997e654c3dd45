#!/usr/bin/env python3
"""Static instruction counts and register use of the fused kernels in gfx950 assembly files.
usage: tools/isa_counts.py file.s [file.s ...]"""
import re, sys
for path in sys.argv[1:]:
    txt = open(path).read()
    lines = txt.split("\n")
    name = None
    cnt = {}
    for l in lines:
        m = re.match(r"^(_Z\w+):", l)
        if m:
            name = m.group(1)
            cnt[name] = [0, 0, 0, 0]
            continue
        if name is None:
            continue
        t = l.strip()
        if "s_endpgm" in t:
            name = None
            continue
        if not t or t[0] in ";." or t.endswith(":"):
            continue
        op = t.split()[0]
        cnt[name][0 if op.startswith("v_") else 1 if op.startswith("s_") else 2 if op.startswith("ds_") else 3] += 1
    regs = {}
    for m in re.finditer(r"\.name:\s+(\S+)\n", txt):
        blk = txt[max(0, m.start() - 2500):m.start() + 2500]
        v = re.search(r"\.vgpr_count:\s+(\d+)", blk)
        sp = re.search(r"\.vgpr_spill_count:\s+(\d+)", blk)
        regs[m.group(1)] = (v.group(1) if v else "?", sp.group(1) if sp else "?")
    print(path)
    for k, v in cnt.items():
        if any(w in k for w in ("k_bounce", "k_primary")):
            short = re.sub(r"^_ZN3ptk\d+\w+?_GLOBAL__N_1\d+", "", k)[:22]
            print(f"  {short:24s} VALU {v[0]:5d} SALU {v[1]:5d} LDS {v[2]:4d} VMEM {v[3]:4d}  vgpr/spill {regs.get(k)}")
