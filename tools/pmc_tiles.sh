#!/bin/bash
# One SQ counter pass of tools/small_tiles.py for rank 0's tile of N (per-kernel means of the whole batches: instructions, wave
# cycles, busy cycles -> residency).  usage: tools/pmc_tiles.sh OUT N
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$1; N=$2; mkdir -p $O
(cd /tmp && export TMPDIR=/tmp && timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_WAIT_ANY --output-format csv -d $O/w$N -o p -- python3 $R/tools/small_tiles.py $N 0 > $O/w$N.log 2>&1)
python3 - $O/w$N/p_counter_collection.csv <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    n = r["Kernel_Name"]
    k = "k_paths" if "k_paths" in n else "k_primary" if "k_primary" in n else "k_collect" if "k_collect" in n else None
    if k: acc[k][r["Counter_Name"]].append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
for k, c in acc.items():
    # whole batches only: dispatches whose wave cycles are within 20 % of the largest
    wc = dict(c["SQ_WAVE_CYCLES"]); top = max(wc.values()); keep = {d for d, v in wc.items() if v > 0.8 * top}
    m = {name: sum(v for d, v in vals if d in keep) / max(1, len([1 for d, v in vals if d in keep])) for name, vals in c.items()}
    print(f"{k}: {len(keep)} whole batches; VALU {m['SQ_INSTS_VALU']/1e6:.1f} M SALU {m['SQ_INSTS_SALU']/1e6:.1f} M wave cycles {m['SQ_WAVE_CYCLES']/1e6:.1f} M busy/32 {m['SQ_BUSY_CYCLES']/32/1e6:.3f} M waves {m['SQ_WAVES']:.0f} residency {m['SQ_WAVE_CYCLES']*4/m['SQ_WAVES']/(m['SQ_BUSY_CYCLES']/32):.2f} wait {m['SQ_WAIT_ANY']/m['SQ_WAVE_CYCLES']:.2f}")
PY
