#!/usr/bin/env python3
"""HBM bytes per launch of the dominant kernel from the FETCH_SIZE / WRITE_SIZE PMC passes of
tools/pmc_passes.sh, corrected as MI355X_MICROARCH.md §HBM prescribes for gfx950: FETCH_SIZE counts
128-B read requests at 64 B, so read bytes = 2 x FETCH_SIZE (calibrated on this workload's own access
pattern: unfused depth-0 k_intersect reads 24 B/ray exactly → ratio 1.985); WRITE_SIZE is exact.
Units are KiB per dispatch.  Writes profiles/dominant_kernel_traffic.json.
usage: tools/pmc_traffic.py PMC_DIR KERNEL_SUBSTR [steady_depths_only]"""
import csv, glob, json, os, sys
d, kern = sys.argv[1], sys.argv[2]
vals = {"FETCH_SIZE": [], "WRITE_SIZE": []}
for f in sorted(glob.glob(os.path.join(d, "pass*", "p_counter_collection.csv"))):
    per = {}
    for r in csv.DictReader(open(f)):
        if kern in r["Kernel_Name"] and r["Counter_Name"] in vals:
            per.setdefault(int(r["Dispatch_Id"]), 0.0)
            per[int(r["Dispatch_Id"])] += float(r["Counter_Value"])
            name = r["Counter_Name"]
    if per:
        vals[name] += list(per.values())
n = min(len(vals["FETCH_SIZE"]), len(vals["WRITE_SIZE"]))
fetch = sum(vals["FETCH_SIZE"]) / len(vals["FETCH_SIZE"]) * 1024
write = sum(vals["WRITE_SIZE"]) / len(vals["WRITE_SIZE"]) * 1024
out = {"kernel": kern, "dispatches": n, "fetch_size_bytes_raw": round(fetch), "write_size_bytes": round(write),
       "hbm_bytes_per_launch": round(2 * fetch + write),
       "note": "2 x FETCH_SIZE + WRITE_SIZE, averaged over all dispatches of the kernel in the PMC run (same bench "
               "configuration as the timed run); gfx950 FETCH_SIZE counts 128-B requests as 64 B"}
print(json.dumps(out, indent=1))
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
json.dump(out, open(os.path.join(ROOT, "profiles", "dominant_kernel_traffic.json"), "w"), indent=1)
