#!/usr/bin/env python3
"""bench.py — headline benchmark of the MI355X wavefront path tracer.

Metric (BASELINE.json): Msamples/s (+ PSNR vs the 5000-spp image), cornell.txt 1920x1080 depth 8.
  step      = one iteration (1 sample per pixel over the whole 1920x1080 frame, all bounces);
  --steps K = K iterations accumulated (default 5000 = BASELINE configs[2]);
  value     = W*H*K / wall seconds of the render loop incl. compaction, gather and the final
              image gather, scene already resident on the GPU.
N GPUs: one process per GPU (torch.distributed, backend nccl == RCCL); the framebuffer is cut into
row-interleaved tiles with GLOBAL pixel indices (so every sample is the same sample as on one GPU), no
collective on the data path, one RCCL gather of the float tiles at image write-out.  (The C++ single-process
form of the same partition is pt_group_* / `pt_render --gpus K`.)

Arithmetic mode (--arith, PT_ARITH_* in include/pt_amd.h): the timed run uses `fast` by default — same
algorithm, draws and decisions as the reference, f32 throughout, held to the stated tolerance against the
reference semantics by tests/test_gpu_arith.py; `modes` reports exact (bit-identical to the oracle) and fma
from the same process.

Extra objects on the JSON line:
  roofline       dominant kernel k_paths (computeIntersections + shadeAndExtendRays + compaction of ALL depths >= 1 in one
                 launch: persistent lanes, no path state through HBM after depth 0): algorithmic bytes = 40 B read per
                 depth-1 ray (its path record, once) + 16 B written per path retired by this kernel (its record, once)
                 / HIP-event time of those launches (events recorded by the library on its own render stream inside the
                 timed region) vs 8 TB/s; `traffic` = HBM bytes per launch from the committed PMC passes
                 (profiles/dominant_kernel_traffic.json holds bytes PER PATH for this arithmetic mode) scaled by this
                 run's paths per launch.  The kernel is NOT bound by HBM (frac ~ 0.13): see roofline_valu.
  roofline_valu  the kernel's instruction-issue ceiling: its dynamic VALU mix (committed PMC passes, SQ_INSTS_VALU_*
                 per 64 rays) priced at the MEASURED issue rates of tools/ubench_valu.hip (profiles/
                 r03_ubench_valu.txt: f32 add / mul / fma 2.3-2.45 SIMD cycles per wave64 instruction, selects /
                 compares / min / max / integer multiplies 4.2, transcendentals 8.2) = needed_cycles, against the SIMD
                 cycles this run spent per 64 rays = spent_cycles.  Unclassified instructions are priced at the full rate,
                 so `frac` is a lower bound.
  pipeline       the whole batch (k_primary + k_paths + k_collect): algorithmic HBM bytes per sample, PMC-measured bytes per
                 sample (profiles/dominant_kernel_traffic.json "pipeline"), and the rate at which this run moved them vs 8 TB/s.
  cpu_baseline   the oracle (kind "port": oracle/pt_oracle.cpp, reference-literal loop, libm math)
                 timed on ONE host thread on a bounded sample of the same workload (rank 0, N=1 only).
  psnr           (N=1 only, outside the timed region) PSNR of 16/64/256/1000-spp prefixes against a 5000-spp
                 GPU image; on a row subset the same PSNRs for the reference semantics (oracle, LIBM) against
                 ITS 5000-spp rows, and their difference (north_star: within 0.1 dB).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np

W, H, DEPTH = 1920, 1080, 8
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8 TB/s spec
ISECT_BYTES_PER_RAY = 56       # SURVEY.md §8(d), unfused computeIntersections
REF_SPP = 5000                 # BASELINE: PSNR vs the 5000-spp image
SIMDS_PER_CU = 4


def psnr(a: np.ndarray, b: np.ndarray) -> float:
    mse = float(np.mean((a.astype(np.float64) - b.astype(np.float64)) ** 2))
    return float("inf") if mse <= 0 else 10.0 * np.log10(1.0 / mse)


def cpu_baseline(scene_path: str, seconds_target: float = 15.0) -> dict:
    """Reference-literal loop (all paths at all depths), glibc math, one thread."""
    from oracle import binding as ob
    ob.set_math_mode(ob.LIBM)
    ob.load_scene(scene_path, res=(W, H))
    rows = 54  # every 20th row: a representative 5 % of the frame
    idx = [r * 20 for r in range(rows)]
    # calibrate with one row, then size the sample to ~seconds_target
    t0 = time.perf_counter()
    ob.render(1, 1, depth=DEPTH, variant=ob.LITERAL, nthreads=1, pix_begin=540 * W, pix_count=W)
    per_row_iter = max(time.perf_counter() - t0, 1e-4)
    spp = max(1, int(seconds_target / (per_row_iter * rows)))
    t0 = time.perf_counter()
    for r in idx:
        ob.render(1, spp, depth=DEPTH, variant=ob.LITERAL, nthreads=1, pix_begin=r * W, pix_count=W)
    dt = time.perf_counter() - t0
    return {"value": round(rows * W * spp / dt / 1e6, 4), "unit": "Msamples/s", "cores": 1, "kind": "port",
            "sample": f"cornell 1920x1080 depth 8: {rows} full rows (every 20th) x {spp} spp, "
                      f"reference-literal loop, {dt:.1f} s on 1 of {os.cpu_count()} host threads"}


PATH_BYTES = 40 + 16           # k_paths: the depth-1 record of a path read once, its retirement record written once


def bounce_accounting(st):
    """Algorithmic bytes, rays and paths of the timed k_paths launches from the renderer's live-ray statistics."""
    live = np.array(st.live_rays[:DEPTH], dtype=np.float64)
    return live, float(PATH_BYTES * live[1]), float(live[1:].sum()), float(live[1])


def pipeline_bytes_per_sample(live, samples, iters_per_batch):
    """Algorithmic HBM bytes per sample of a whole batch: depth 0 writes a 40-B record per survivor or a 16-B record per retiree;
    k_paths reads the former and writes a 16-B record per path; k_collect reads every record and reads + writes the image
    (24 B per pixel and batch)."""
    s1 = live[1] / samples
    return 40 * s1 + 16 * (1 - s1) + PATH_BYTES * s1 + 16 + 24.0 / max(1, iters_per_batch)


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5000)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--arith", choices=["exact", "fma", "fast"], default="fast", help="arithmetic mode of the timed run (PT_ARITH_*)")
    ap.add_argument("--iters-per-batch", type=int, default=0)
    ap.add_argument("--queues", type=int, default=0)
    ap.add_argument("--blocks-per-cu", type=int, default=0)
    ap.add_argument("--legacy-traversal", action="store_true", help="A/B: per-lane BVH walk kernel")
    ap.add_argument("--unfused-primary", action="store_true", help="A/B: depth 0 as separate generate/intersect/shade launches")
    ap.add_argument("--contiguous-tiles", action="store_true", help="A/B: one block of rows per rank instead of interleaved rows")
    ap.add_argument("--unfused-bounces", action="store_true", help="A/B: depths >= 1 as separate intersect + shade launches")
    ap.add_argument("--debug-flags", type=int, default=0, help="A/B switches (16 / 32: result-neutral; 1-8 need a -DPT_ABLATE library)")
    ap.add_argument("--no-kernel-events", action="store_true", help="do not bracket the dominant kernel's launches with HIP events")
    ap.add_argument("--no-extras", action="store_true", help="skip modes, cpu_baseline and psnr (N=1 extras)")
    ap.add_argument("--stress", action="store_true", help="second JSON line: BASELINE config C5 (10,170 primitives, 1080p)")
    ap.add_argument("--save", type=str, default="", help="write the final image as PREFIX.png/.pfm")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    from cosc_4397_pathtracing_raytracing_project_amd import capi, parallel, scenes

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("bench.py --gpus N>1 must be launched with torch.distributed.run (one rank per GPU)")
        args.gpus = world
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    # Rehearsal on a 1-GPU box: PT_DIST_BACKEND=gloo lets several ranks share one card (RCCL needs one GPU per
    # rank); the tiles are then gathered through host memory.  The driver's runs use the default, nccl.
    backend = os.environ.get("PT_DIST_BACKEND", "nccl")
    ndev = torch.cuda.device_count()
    if backend == "nccl" and local_rank >= ndev:
        raise SystemExit(f"rank {rank}: local rank {local_rank} but only {ndev} GPU(s) visible")
    local_rank = local_rank % max(1, ndev)
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)

    tmp = tempfile.mkdtemp(prefix="ptbench_")
    scene_path = scenes.write_scene(scenes.cornell_scene_text(res=(W, H), iterations=args.steps, depth=DEPTH),
                                    os.path.join(tmp, f"cornell_{rank}.txt"))
    scene = capi.Scene(scene_path, res=(W, H))
    striped = world > 1 and not args.contiguous_tiles
    if striped:
        topt = parallel.striped_tile_for_rank(W, H, rank, world)
    else:
        begin, count = parallel.tile_for_rank(W, H, rank, world)
        topt = dict(pixel_begin=begin, pixel_count=count)
    count = topt["pixel_count"]
    tile = torch.zeros((count, 3), dtype=torch.float32, device=f"cuda:{local_rank}")

    def make_renderer(time_kernels: bool, arith: str = args.arith, **over):
        kw = dict(device=local_rank, **topt, iters_per_batch=args.iters_per_batch, num_queues=args.queues,
                  blocks_per_cu=args.blocks_per_cu, time_kernels=time_kernels, legacy_traversal=args.legacy_traversal,
                  debug_flags=args.debug_flags, arith=arith, unfused_primary=args.unfused_primary,
                  unfused_bounces=args.unfused_bounces)
        kw.update(over)
        return capi.Renderer(scene, **kw)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    r = make_renderer(not args.no_kernel_events)
    # warm-up: untimed iterations, then the accumulation is restarted on the same (now touched) buffers
    if args.warmup > 0:
        r.render(1, args.warmup)
        r.readback_device(tile.data_ptr())
        parallel.gather_tiles(tile, W, H, rank, world, striped=striped)
        r.clear()

    barrier()
    t0 = time.perf_counter()
    r.render(1, args.steps)                 # K steps: all launches are asynchronous on the render stream
    r.readback_device(tile.data_ptr())      # waits for the stream, tile SUM image → torch tensor
    full = parallel.gather_tiles(tile, W, H, rank, world, striped=striped)  # single RCCL gather at image write-out
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device=f"cuda:{local_rank}" if backend == "nccl" else "cpu")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())

    st = r.stats()
    live, alg_bytes, units, paths = bounce_accounting(st)
    isect_s = st.intersect_ms / 1e3
    roofline = roofline_valu = None
    prof_all = {}
    ppath = os.path.join(ROOT, "profiles", "dominant_kernel_traffic.json")
    if os.path.exists(ppath):
        try:
            prof_all = json.load(open(ppath))
        except Exception:
            prof_all = {}
    prof = prof_all.get(args.arith, {})
    if st.intersect_launches > 0 and isect_s > 0:
        timed = live[1:] if st.primary_fused else live  # depths covered by the timed launches
        if st.bounces_fused:
            kernel = "k_paths (computeIntersections + shadeAndExtendRays + compaction, depths 1..7 in one launch)"
            per_unit = "40 B read per depth-1 ray (its path record, once) + 16 B written per path (its retirement record, once)"
        else:
            kernel = "k_intersect (computeIntersections)"
            alg_bytes = float(ISECT_BYTES_PER_RAY * timed.sum())
            units = float(timed.sum())
            per_unit = "56 B per live ray (24 read + 32 written)"
        achieved = alg_bytes / isect_s / 1e9
        rays_per_launch = units / st.intersect_launches
        paths_per_launch = paths / st.intersect_launches
        avg_us = isect_s * 1e6 / st.intersect_launches
        same_kernel = prof.get("kernel", "").split(" ")[0] == kernel.split(" ")[0]
        traffic = round(prof["hbm_bytes_per_path"] * paths_per_launch) if same_kernel and "hbm_bytes_per_path" in prof else None
        roofline = {"bound": "hbm", "kernel": kernel, "achieved": round(achieved, 2),
                    "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                    "traffic_source": (f"profiles/dominant_kernel_traffic.json [{args.arith}]: {prof['hbm_bytes_per_path']:.1f} B/path "
                                       f"(2 x FETCH_SIZE + WRITE_SIZE) x this run's paths per launch") if traffic else None,
                    "launches": int(st.intersect_launches), "avg_launch_us": round(avg_us, 3),
                    "algorithmic_bytes_per_launch": round(alg_bytes / st.intersect_launches, 1),
                    "algorithmic_bytes_per_unit": per_unit,
                    "rays_per_launch": round(rays_per_launch, 1), "paths_per_launch": round(paths_per_launch, 1),
                    "depths_timed": "1..7 (depth 0 runs in the fused primary kernel)" if st.primary_fused else "0..7",
                    "live_rays_per_sample": round(float(live.sum()) / max(1, st.samples), 4),
                    "note": "the fused kernel keeps path state on chip: it is bound by instruction issue, not by HBM (roofline_valu); "
                            "SURVEY 8(d)'s 56 B per live ray describes the unfused k_intersect (bench.py --unfused-bounces)"}
        roofline["frac_56B"] = round(ISECT_BYTES_PER_RAY * units / isect_s / 1e9 / HBM_PEAK_GBS, 4)  # as if every traced ray moved SURVEY 8(d)'s 56 B
        if same_kernel and "valu_ceiling_simd_cycles_per_group" in prof:
            # instruction-issue ceiling: SIMD cycles the measured mix needs per 64 rays vs SIMD cycles spent per 64 rays
            groups_per_simd = rays_per_launch / 64.0 / (st.num_cus * SIMDS_PER_CU)
            clock = (prof.get("shader_clock_ghz") or 2.1) * 1e9
            spent = avg_us * 1e-6 * clock / groups_per_simd
            need = prof["valu_ceiling_simd_cycles_per_group"]
            roofline_valu = {"bound": "valu", "kernel": kernel.split(" ")[0],
                             "valu_per_64_rays": prof["valu_per_group"], "salu_per_64_rays": prof.get("salu_per_group"),
                             "valu_mix_per_64_rays": prof.get("valu_mix_per_group"),
                             "needed_cycles": round(need, 1), "spent_cycles": round(spent, 1),
                             "unit": "SIMD cycles per 64 rays (needed by the VALU mix at measured issue rates / spent)",
                             "frac": round(need / spent, 4), "shader_clock_ghz_pmc": prof.get("shader_clock_ghz"),
                             "source": f"profiles/dominant_kernel_traffic.json [{args.arith}] (PMC: SQ_INSTS_VALU_* per 64 rays) priced with "
                                       f"profiles/r03_ubench_valu.txt; unclassified instructions at the full rate: frac is a lower bound"}

    samples = float(W) * H * args.steps
    out = {
        "metric": "Msamples/s, cornell.txt 1920x1080 depth 8 (+PSNR vs 5000spp ref)",
        "value": round(samples / dt / 1e6, 3),
        "unit": "Msamples/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(dt * 1e3 / args.steps, 6),
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": f"cornell.txt 1920x1080, {args.steps} spp, depth 8, BVH + compaction, "
                               f"{world}x MI355X " + ("interleaved-row tiles" if striped else "row tiles") + (", one RCCL gather" if world > 1 else ""),
                   "arith": args.arith, "reference_equivalent_arith": "fma",
                   "iters_per_batch": int(st.iters_per_batch), "queues": int(st.num_queues),
                   "grid_blocks": int(st.grid_blocks), "cus": int(st.num_cus),
                   "device_mem_mb": round(st.device_bytes / 2 ** 20, 1),
                   "kernel_events": not args.no_kernel_events,
                   "traversal": "legacy per-lane" if args.legacy_traversal else "wave-cooperative"},
        "roofline": roofline,
        "roofline_valu": roofline_valu,
    }
    if st.bounces_fused and st.primary_fused and st.samples > 0:
        alg_ps = pipeline_bytes_per_sample(live, float(st.samples), int(st.iters_per_batch))
        pmc_ps = prof.get("pipeline_hbm_bytes_per_sample")
        out["pipeline"] = {"kernels": "k_primary + k_paths + k_collect (+ k_count_stats)", "algorithmic_bytes_per_sample": round(alg_ps, 2),
                           "pmc_bytes_per_sample": pmc_ps, "pmc_source": "profiles/dominant_kernel_traffic.json [%s] pipeline_hbm_bytes_per_sample "
                           "(2 x FETCH_SIZE + WRITE_SIZE summed over the batch's kernels / samples)" % args.arith if pmc_ps else None,
                           "achieved": round(alg_ps * samples / dt / 1e9, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                           "frac": round(alg_ps * samples / dt / 1e9 / HBM_PEAK_GBS, 4),
                           "frac_pmc": round(pmc_ps * samples / dt / 1e9 / HBM_PEAK_GBS, 4) if pmc_ps else None}

    if rank == 0 and world == 1 and not args.no_extras:
        img = full.cpu().numpy()
        r.free()
        # ---- all three arithmetic modes measured ALIKE, same process, same workload: a renderer of its own, 200 untimed
        # steps, accumulation restarted, <= 1000 timed steps (`steady_value`).  The headline `value` stays what the
        # driver's own --steps / --warmup measured; modes[headline].value repeats it.
        modes = {}
        msteps = min(max(args.steps, 200), 1000)
        for m in ("exact", "fma", "fast"):
            rr = make_renderer(True, arith=m)
            rr.render(1, 200)  # untimed; long enough for the memory the previous leg's renderer released to settle (LABNOTES.md section 6)
            rr.readback_device(tile.data_ptr())
            rr.clear()  # as for the headline run: warm-up and timed steps on the same buffers
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            rr.render(1, msteps)
            rr.readback_device(tile.data_ptr())
            torch.cuda.synchronize()
            d1 = time.perf_counter() - t1
            s1 = rr.stats()
            _, a1, u1, _p1 = bounce_accounting(s1)
            sv = round(W * H * msteps / d1 / 1e6, 3)
            modes[m] = {"value": out["value"] if m == args.arith else sv, "steps": args.steps if m == args.arith else msteps,
                        "steady_value": sv, "steady_steps": msteps,
                        "k_paths_us": round(s1.intersect_ms * 1e3 / max(1, s1.intersect_launches), 3),
                        "rays_per_launch": round(u1 / max(1, s1.intersect_launches), 1),
                        "hbm_frac": round(a1 / (s1.intersect_ms / 1e3) / 1e9 / HBM_PEAK_GBS, 4) if s1.intersect_ms > 0 else None,
                        "hbm_frac_56B": round(ISECT_BYTES_PER_RAY * u1 / (s1.intersect_ms / 1e3) / 1e9 / HBM_PEAK_GBS, 4) if s1.intersect_ms > 0 else None}
            rr.free()
        out["modes"] = modes
        # the reference builds WITHOUT -use_fast_math (CMakeLists.txt:26-30): FMA contraction, IEEE divide / sqrt — that is
        # the `fma` mode; quote it beside the headline
        out["value_fma"] = modes["fma"]["steady_value"]
        out["value_exact"] = modes["exact"]["steady_value"]

        # ---- PSNR vs the 5000-spp image (always 5000, whatever --steps), outside the timed region ----
        def render_sum(n, arith=args.arith):
            rr = capi.Renderer(scene, device=local_rank, arith=arith)
            rr.render(1, n)
            a = rr.readback()
            rr.free()
            return a
        ref = img if args.steps == REF_SPP else render_sum(REF_SPP)
        ref_avg = ref / np.float32(REF_SPP)
        ps = {}
        prefixes = {}
        for n in (16, 64, 256, 1000):
            prefixes[n] = render_sum(n)
            ps[f"{n}spp_vs_{REF_SPP}spp"] = round(psnr(prefixes[n] / np.float32(n), ref_avg), 2)
        # ---- the same against the reference semantics (oracle, LIBM math, reference-literal loop) on whole rows ----
        from oracle import binding as ob
        rows = [540, 800]
        threads = min(32, os.cpu_count() or 1)
        ob.set_math_mode(ob.LIBM)
        ob.load_scene(scene_path, res=(W, H))

        def oracle_rows(n):
            return np.concatenate([ob.render(1, n, depth=DEPTH, variant=ob.LITERAL, nthreads=threads, pix_begin=row * W, pix_count=W)
                                   for row in rows])

        def gpu_rows(a):
            return np.concatenate([a[row * W:(row + 1) * W] for row in rows])
        o_ref = oracle_rows(REF_SPP) / np.float32(REF_SPP)
        g_ref = gpu_rows(ref) / np.float32(REF_SPP)
        ps["rows_checked"] = rows
        ps[f"gpu_{REF_SPP}spp_vs_reference_semantics_{REF_SPP}spp_db"] = round(psnr(g_ref, o_ref), 2)
        delta = {}
        for n in (16, 64, 256, 1000):
            p_gpu = psnr(gpu_rows(prefixes[n]) / np.float32(n), o_ref)
            p_orc = psnr(oracle_rows(n) / np.float32(n), o_ref)
            delta[f"{n}spp"] = {"gpu_db": round(p_gpu, 3), "reference_semantics_db": round(p_orc, 3), "delta_db": round(abs(p_gpu - p_orc), 4)}
        ps["psnr_delta_vs_oracle_db"] = delta
        ps["max_delta_db"] = max(v["delta_db"] for v in delta.values())
        ps["within_0.1_db"] = bool(ps["max_delta_db"] <= 0.1)
        if args.arith == "exact":
            ob.set_math_mode(ob.PORTABLE)
            ob.load_scene(scene_path, res=(W, H))
            exact_rows = np.concatenate([ob.render(1, args.steps, depth=DEPTH, variant=ob.RETIRE, nthreads=threads, pix_begin=row * W, pix_count=W) for row in rows])
            ps["gpu_bit_exact_vs_oracle_portable"] = bool(np.array_equal(gpu_rows(img).view(np.uint32), exact_rows.view(np.uint32)))
        out["psnr"] = ps
        assert ps["within_0.1_db"], f"PSNR differs from the reference semantics by {ps['max_delta_db']} dB (> 0.1)"
        out["cpu_baseline"] = cpu_baseline(scene_path)
        if args.save:
            capi.save_png(args.save + ".png", img, W, H, float(args.steps))
            capi.save_pfm(args.save + ".pfm", img, W, H, float(args.steps))
    else:
        r.free()

    if rank == 0:
        print(json.dumps(out), flush=True)

    if args.stress and rank == 0 and world == 1:
        # optional second line: BASELINE config C5 (tools/run_config.py stress), 200 spp by default
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import run_config
        print(json.dumps(run_config.run("stress", spp=200, arith=args.arith)), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
