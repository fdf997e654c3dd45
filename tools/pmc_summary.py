#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc passes (tools/pmc_passes.sh) per kernel, and per depth for the
intersect / shade kernels (dispatch order inside each batch)."""
import csv, collections, glob, os, sys

def load(d):
    out = collections.defaultdict(dict)   # dispatch id -> {counter: value, 'name':..}
    for f in sorted(glob.glob(os.path.join(d, "pass*", "p_counter_collection.csv"))):
        p = os.path.basename(os.path.dirname(f))
        for r in csv.DictReader(open(f)):
            key = (p, int(r["Dispatch_Id"]))
            out[key]["name"] = r["Kernel_Name"]
            out[key][r["Counter_Name"]] = out[key].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
            out[key]["grid"] = int(r.get("Grid_Size", 0) or 0)
    return out

def short(n):
    for k in ("k_intersect_legacy", "k_intersect", "k_shade", "k_generate", "k_primary", "k_paths", "k_bounce_all", "k_bounce", "k_gather", "k_collect", "k_count_stats"):
        if k in n:
            return k
    return None

def main():
    d = sys.argv[1]
    data = load(d)
    # group by pass → ordered dispatches → assign depth by order within batch
    per = collections.defaultdict(lambda: collections.defaultdict(list))
    bypass = collections.defaultdict(list)
    for (p, did), v in sorted(data.items()):
        bypass[p].append((did, v))
    for p, lst in bypass.items():
        depth = {"k_intersect": 0, "k_shade": 0, "k_intersect_legacy": 0, "k_bounce": 1}
        for did, v in lst:
            s = short(v["name"])
            if s is None:
                continue
            if s in ("k_generate", "k_primary"):
                depth = {k: (1 if k == "k_bounce" else 0) for k in depth}
            tag = s
            if s in depth:
                tag = f"{s}[d{depth[s]}]"
                depth[s] += 1
            for c, val in v.items():
                if c in ("name", "grid"):
                    continue
                per[tag][c].append(val)
                per[s + "[all]"][c].append(val) if s in depth else None
    for tag in sorted(per):
        row = per[tag]
        n = max(len(x) for x in row.values())
        print(f"== {tag}  ({n} dispatches)")
        for c in sorted(row):
            vals = row[c]
            print(f"   {c:28s} mean {sum(vals)/len(vals):16.1f}")
if __name__ == "__main__":
    main()
