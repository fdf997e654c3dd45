#!/usr/bin/env python3
"""Per-mode figures of the dominant kernel from the PMC passes of tools/pmc_passes.sh, written to
profiles/dominant_kernel_traffic.json[MODE] for bench.py (roofline.traffic, roofline_valu):
  hbm_bytes_per_path  (2 x FETCH_SIZE + WRITE_SIZE) / depth-1 paths per launch (k_paths moves a path's bytes once, however many
                      rays it traces; hbm_bytes_per_ray = the same per traced ray).  Corrected as MI355X_MICROARCH.md §HBM prescribes for
                      gfx950: FETCH_SIZE counts 128-B read requests at 64 B (calibrated on this workload: unfused depth-0
                      k_intersect reads exactly 24 B/ray -> ratio 1.985); WRITE_SIZE is exact.  Units: KiB per dispatch.
  pipeline_hbm_bytes_per_sample
                      the same sum over ALL kernels of a batch (k_primary, k_paths, k_collect, k_count_stats) / samples per batch
  valu_per_group      SQ_INSTS_VALU / (rays per launch / 64) (salu_per_group likewise)
  valu_busy_frac      SQ_ACTIVE_INST_VALU (quad-cycles, summed over waves) x 4 / (SIMDs x kernel cycles), kernel cycles =
                      SQ_BUSY_CYCLES / shader engines
  shader_clock_ghz    kernel cycles / kernel duration from the kernel trace of the same pass
  valu_mix_per_group  the dynamic instruction mix (SQ_INSTS_VALU_ADD_F32 / MUL_F32 / FMA_F32 / TRANS_F32 / INT32 / INT64 / CVT,
                      the rest = moves, selects, compares, min / max, lane ops) per 64-ray group
  valu_ceiling_simd_cycles_per_group
                      what that mix costs at the MEASURED issue rates of profiles/r03_ubench_valu.txt (tools/ubench_valu.hip,
                      SIMD cycles per wave64 instruction at 4 waves per SIMD): f32 add / mul / fma 2.3-2.45, transcendental 8.2,
                      64-bit integer and conversions 4.2; 32-bit integer and "the rest" are mixtures of full-rate (add, xor, mov: 2.3)
                      and half-rate (multiply, shift-add, select, compare, min / max, lane ops: 4.2) instructions and are priced
                      at the FULL rate, so this is a lower bound on the cycles — roofline_valu.frac in bench.py is therefore a
                      lower bound on how close the kernel runs to its instruction-issue ceiling.
The rays per launch come from the bench line the first pass printed (pass1.json).
usage: tools/pmc_traffic.py PMC_DIR KERNEL_SUBSTR MODE"""
import csv, glob, json, os, sys
d, kern, mode = sys.argv[1], sys.argv[2], sys.argv[3]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
vals = {}
dur = []
for f in sorted(glob.glob(os.path.join(d, "pass*", "p_counter_collection.csv"))):
    per = {}
    for r in csv.DictReader(open(f)):
        if kern in r["Kernel_Name"]:
            per.setdefault(r["Counter_Name"], {}).setdefault(int(r["Dispatch_Id"]), 0.0)
            per[r["Counter_Name"]][int(r["Dispatch_Id"])] += float(r["Counter_Value"])
    for c, v in per.items():
        vals.setdefault(c, []).extend(v.values())
    kt = os.path.join(os.path.dirname(f), "p_kernel_trace.csv")
    if os.path.exists(kt) and "SQ_BUSY_CYCLES" in per:
        for r in csv.DictReader(open(kt)):
            if kern in r["Kernel_Name"]:
                dur.append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
mean = {c: sum(v) / len(v) for c, v in vals.items()}


def ubench_costs():
    """SIMD cycles per wave64 instruction at 4 resident waves per SIMD, from the committed micro-benchmark output."""
    cost = {}
    path = os.path.join(ROOT, "profiles", "r03_ubench_valu.txt")
    if os.path.exists(path):
        for l in open(path):
            if "|" in l and not l.startswith(("#", "instruction")):
                name, _, simd = [x.strip() for x in l.split("|")[:3]]
                try:
                    cost[name.split()[0]] = float(simd.split()[2])  # column W = 4
                except (ValueError, IndexError):
                    pass
    return cost


def valu_mix(mean, rays):
    if "SQ_INSTS_VALU_FMA_F32" not in mean:
        return {}
    g = rays / 64.0
    cost = ubench_costs()
    full = cost.get("v_mul_f32", 2.3)
    cls = {"add_f32": ("SQ_INSTS_VALU_ADD_F32", cost.get("v_add_f32", 2.3)), "mul_f32": ("SQ_INSTS_VALU_MUL_F32", cost.get("v_mul_f32", 2.3)),
           "fma_f32": ("SQ_INSTS_VALU_FMA_F32", cost.get("v_fma_f32", 2.45)), "trans_f32": ("SQ_INSTS_VALU_TRANS_F32", cost.get("v_rcp_f32", 8.2)),
           "int32": ("SQ_INSTS_VALU_INT32", full), "int64": ("SQ_INSTS_VALU_INT64", cost.get("v_lshl_add_u64", 4.25)),
           "cvt": ("SQ_INSTS_VALU_CVT", cost.get("v_cvt_f32_u32", 4.15))}
    mix, cycles, known = {}, 0.0, 0.0
    for k, (ctr, c) in cls.items():
        n = mean.get(ctr, 0.0) / g
        mix[k] = round(n, 1)
        cycles += n * c
        known += n
    other = mean["SQ_INSTS_VALU"] / g - known
    mix["other_mov_select_compare_minmax_lane"] = round(other, 1)
    cycles += other * full
    return {"valu_mix_per_group": mix, "valu_ceiling_simd_cycles_per_group": round(cycles, 1),
            "valu_ceiling_source": "profiles/r03_ubench_valu.txt (SIMD cycles per wave64 instruction at W = 4); int32 and 'other' priced at the full rate: a lower bound"}

line = json.loads(open(os.path.join(d, "pass1.json")).read().strip().split("\n")[-1])
rays = line["roofline"]["rays_per_launch"] if line.get("roofline") else None
paths = line["roofline"].get("paths_per_launch", rays) if line.get("roofline") else None
if rays is None:  # passes run with --no-kernel-events: derive from the statistics in config
    raise SystemExit("pass1.json has no roofline object; run the passes with kernel events enabled")
cus = line["config"]["cus"]
fetch, write = mean["FETCH_SIZE"] * 1024, mean["WRITE_SIZE"] * 1024
cycles = mean["SQ_BUSY_CYCLES"] / 32.0  # 32 shader engines report
out = {"kernel": kern, "dispatches": len(vals["FETCH_SIZE"]), "rays_per_launch": rays,
       "fetch_size_bytes_raw": round(fetch), "write_size_bytes": round(write),
       "paths_per_launch": paths,
       "hbm_bytes_per_launch": round(2 * fetch + write), "hbm_bytes_per_ray": round((2 * fetch + write) / rays, 2),
       "hbm_bytes_per_path": round((2 * fetch + write) / paths, 2),
       "algorithmic_bytes_per_ray": round(line["roofline"]["algorithmic_bytes_per_launch"] / rays, 2),
       "algorithmic_bytes_per_path": round(line["roofline"]["algorithmic_bytes_per_launch"] / paths, 2),
       "valu_per_group": round(mean["SQ_INSTS_VALU"] / (rays / 64), 1), "salu_per_group": round(mean["SQ_INSTS_SALU"] / (rays / 64), 1),
       "lds_per_group": round(mean["SQ_INSTS_LDS"] / (rays / 64), 1),
       "valu_busy_frac": round(mean["SQ_ACTIVE_INST_VALU"] * 4 / (cus * 4 * cycles), 4),
       "wave_wait_frac": round(mean["SQ_WAIT_ANY"] / mean["SQ_WAVE_CYCLES"], 4),
       "wave_issue_stall_frac": round(mean["SQ_WAIT_INST_ANY"] / mean["SQ_WAVE_CYCLES"], 4),
       "shader_clock_ghz": round(cycles / (sum(dur) / len(dur)), 3) if dur else None,
       **valu_mix(mean, rays),
       "avg_launch_us_under_pmc": round(sum(dur) / len(dur) / 1e3, 1) if dur else None,
       "note": "PMC passes of `bench.py --arith %s` (tools/pmc_passes.sh); per-ray figures are scaled by a run's own rays per launch in bench.py" % mode}
# whole batch: every kernel's HBM bytes, per sample (FETCH_SIZE / WRITE_SIZE passes; one k_primary dispatch per batch)
tot = {"FETCH_SIZE": 0.0, "WRITE_SIZE": 0.0}
batches = 0
for f in sorted(glob.glob(os.path.join(d, "pass*", "p_counter_collection.csv"))):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] in tot and "ptk::" in r["Kernel_Name"]:
            tot[r["Counter_Name"]] += float(r["Counter_Value"])
            if r["Counter_Name"] == "FETCH_SIZE" and "k_primary" in r["Kernel_Name"]:
                batches += 1
# (dispatches are reported once per XCD instance of the counter: count distinct dispatch ids instead)
ids = set()
for f in sorted(glob.glob(os.path.join(d, "pass*", "p_counter_collection.csv"))):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == "FETCH_SIZE" and "k_primary" in r["Kernel_Name"]:
            ids.add((f, r["Dispatch_Id"]))
batches = len(ids)
if batches:
    k_iters = line["config"]["iters_per_batch"]
    samples_per_batch = 1920 * 1080 * k_iters
    out["pipeline_hbm_bytes_per_sample"] = round((2 * tot["FETCH_SIZE"] + tot["WRITE_SIZE"]) * 1024 / batches / samples_per_batch, 2)
    out["pipeline_batches"] = batches
print(json.dumps(out, indent=1))
path = os.path.join(ROOT, "profiles", "dominant_kernel_traffic.json")
try:
    allm = json.load(open(path))
    if "kernel" in allm:  # round-1 layout
        allm = {}
except Exception:
    allm = {}
allm[mode] = out
json.dump(allm, open(path, "w"), indent=1)
