#!/usr/bin/env python3
"""Per-depth kernel durations from a rocprofv3 --kernel-trace CSV of a bench run."""
import csv, statistics, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
seq, cur = [], None
for r in rows:
    n = r["Kernel_Name"]
    s = "gen" if "k_primary" in n else "isect" if "k_intersect" in n else "shade" if "k_shade" in n else "gen" if "k_generate" in n else "gather" if "k_gather" in n else "other"
    if s == "gen":
        cur = []
        seq.append(cur)
    if cur is not None:
        cur.append((s, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1000, int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
D = max(len([x for x in b if x[0] == "isect"]) for b in seq)
full = [b for b in seq if len([x for x in b if x[0] == "isect"]) == D][2:]
print("first kernel of batch:", "k_primary/k_generate")
ti = ts = 0
for i in range(D):
    di = statistics.median([[x for x in b if x[0] == "isect"][i][1] for b in full])
    ds = statistics.median([[x for x in b if x[0] == "shade"][i][1] for b in full])
    ti += di; ts += ds
    print(f"depth {i}: isect {di:7.1f} us   shade {ds:7.1f} us")
gen = statistics.median([b[0][1] for b in full])
gat = statistics.median([[x for x in b if x[0] == "gather"][0][1] for b in full])
span = statistics.median([b[-1][3] - b[0][2] for b in full]) / 1000
print(f"isect {ti:.1f}  shade {ts:.1f}  gen {gen:.1f}  gather {gat:.1f}  sum {ti+ts+gen+gat:.1f}  batch span {span:.1f} us  ({len(full)} batches)")
