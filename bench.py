#!/usr/bin/env python3
"""bench.py — headline benchmark of the MI355X wavefront path tracer.

Metric (BASELINE.json): Msamples/s (+ PSNR vs the 5000-spp image), cornell.txt 1920x1080 depth 8.
  step      = one iteration (1 sample per pixel over the whole 1920x1080 frame, all bounces);
  --steps K = K iterations accumulated (default 5000 = BASELINE configs[2]);
  value     = W*H*K / wall seconds of the render loop incl. compaction, gather and the final
              image gather, scene already resident on the GPU.
N GPUs: one process per GPU (torch.distributed, backend nccl == RCCL); the framebuffer is cut
into N row tiles with GLOBAL pixel indices (so every sample is the same sample as on one GPU),
no collective on the data path, one RCCL gather of the float tiles at image write-out.

Extra objects on the JSON line:
  roofline     computeIntersections: 56 B per live ray (24 B read o,d + 32 B written t,n,mat,p;
               SURVEY.md §8d) x live rays traced / HIP-event time of those launches (events recorded
               by the library on its own render stream during the timed region) vs 8 TB/s HBM.
  cpu_baseline the oracle (kind "port": oracle/pt_oracle.cpp, reference-literal loop, libm math)
               timed on ONE host thread on a bounded sample of the same workload (rank 0, N=1 only).
  psnr         (N=1 only, outside the timed region) PSNR of 16/64/256/1000-spp prefixes against the
               K-spp image, and the cross-implementation check of a few full-resolution rows
               against the CPU oracle at the full sample count.
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
ISECT_BYTES_PER_RAY = 56


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
    return {"value": rows * W * spp / dt / 1e6, "unit": "Msamples/s", "cores": 1, "kind": "port",
            "sample": f"cornell 1920x1080 depth 8: {rows} full rows (every 20th) x {spp} spp, "
                      f"reference-literal loop, {dt:.1f} s on 1 of {os.cpu_count()} host threads"}


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5000)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--iters-per-batch", type=int, default=0)
    ap.add_argument("--queues", type=int, default=0)
    ap.add_argument("--blocks-per-cu", type=int, default=0)
    ap.add_argument("--legacy-traversal", action="store_true", help="A/B: per-lane BVH walk kernel")
    ap.add_argument("--unfused-primary", action="store_true", help="A/B: depth 0 as separate generate/intersect/shade launches")
    ap.add_argument("--contiguous-tiles", action="store_true", help="A/B: one block of rows per rank instead of interleaved rows")
    ap.add_argument("--unfused-bounces", action="store_true", help="A/B: depths >= 1 as separate intersect + shade launches")
    ap.add_argument("--debug-flags", type=int, default=0, help="A/B switches (16 / 32: result-neutral; 1-8 need a -DPT_ABLATE library)")
    ap.add_argument("--arith", choices=["exact", "fma", "fast"], default="fast", help="arithmetic mode of the kernels (PT_ARITH_*)")
    ap.add_argument("--no-kernel-events", action="store_true", help="do not bracket intersect launches with HIP events")
    ap.add_argument("--no-extras", action="store_true", help="skip cpu_baseline and psnr (N=1 extras)")
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

    def make_renderer(time_kernels: bool):
        return capi.Renderer(scene, device=local_rank, **topt,
                             iters_per_batch=args.iters_per_batch, num_queues=args.queues,
                             blocks_per_cu=args.blocks_per_cu, time_kernels=time_kernels,
                             legacy_traversal=args.legacy_traversal, debug_flags=args.debug_flags, arith=args.arith,
                             unfused_primary=args.unfused_primary, unfused_bounces=args.unfused_bounces)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    r = make_renderer(not args.no_kernel_events)
    # warm-up: untimed iterations on a scratch accumulation (restarted below)
    if args.warmup > 0:
        r.render(1, args.warmup)
        r.readback_device(tile.data_ptr())
        parallel.gather_tiles(tile, W, H, rank, world, striped=striped)
    r.free()
    r = make_renderer(not args.no_kernel_events)

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
    live = np.array(st.live_rays[:DEPTH], dtype=np.float64)
    timed = live[1:] if st.primary_fused else live  # depths covered by the timed computeIntersections launches
    isect_s = st.intersect_ms / 1e3
    if world > 1:  # roofline is per GPU: report rank 0's kernel (all ranks run the same kernel on their tile)
        pass
    roofline = None
    if st.intersect_launches > 0 and isect_s > 0:
        nxt = np.append(live[1:], 0.0)  # survivors of depth d = live rays of depth d+1 (none after the last depth)
        if st.bounces_fused:
            # fused bounce kernel, depths >= 1: 40 B path state read per ray; written: 40 B per survivor or
            # 12 B per retired sample (the hit record of SURVEY §8d's 56 + 104 B never reaches HBM)
            kernel = "k_bounce (computeIntersections + shadeAndExtendRays + compaction, depths 1..7)"
            alg_bytes = float((40 * live[1:] + 40 * nxt[1:] + 12 * (live[1:] - nxt[1:])).sum())
            units = float(live[1:].sum())
            per_unit = "40 B read + 40 B (survivor) / 12 B (retired) written per ray"
        else:
            kernel = "k_intersect (computeIntersections)"
            alg_bytes = float(ISECT_BYTES_PER_RAY * timed.sum())
            units = float(timed.sum())
            per_unit = "56 B per live ray (24 read + 32 written)"
        achieved = alg_bytes / isect_s / 1e9
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "dominant_kernel_traffic.json")
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                if tj.get("kernel", "").split(" ")[0] == kernel.split(" ")[0]:
                    traffic = tj.get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        roofline = {"bound": "hbm", "kernel": kernel, "achieved": round(achieved, 2),
                    "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                    "launches": int(st.intersect_launches), "avg_launch_us": round(isect_s * 1e6 / st.intersect_launches, 3),
                    "algorithmic_bytes_per_launch": round(alg_bytes / st.intersect_launches, 1),
                    "algorithmic_bytes_per_unit": per_unit,
                    "rays_per_launch": round(units / st.intersect_launches, 1),
                    "depths_timed": "1..7 (depth 0 runs in the fused primary kernel)" if st.primary_fused else "0..7",
                    "live_rays_per_sample": round(float(live.sum()) / max(1, st.samples), 4)}

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
                   "iters_per_batch": int(st.iters_per_batch), "queues": int(st.num_queues),
                   "grid_blocks": int(st.grid_blocks), "cus": int(st.num_cus),
                   "device_mem_mb": round(st.device_bytes / 2 ** 20, 1),
                   "kernel_events": not args.no_kernel_events,
                   "traversal": "legacy per-lane" if args.legacy_traversal else "wave-cooperative"},
        "roofline": roofline,
    }
    # Context: the reference pipeline's own algorithmic traffic is 44 + 160*L + 40 B per sample
    # (generate 44, intersect 56 + shade 104 per live segment, gather 40; L = live segments per sample), i.e. a
    # ceiling of 8 TB/s / that figure if every stage streamed its records through HBM.  The fused kernels avoid
    # most of that traffic, so the sample rate is also quoted as the HBM rate that pipeline would have needed.
    L = float(live.sum()) / max(1, st.samples)
    ref_bytes = 44.0 + 160.0 * L + 40.0
    out["pipeline_equivalent"] = {"reference_pipeline_bytes_per_sample": round(ref_bytes, 1),
                                  "equivalent_GBps": round(out["value"] * 1e6 * ref_bytes / 1e9 / max(1, world), 1),
                                  "frac_of_hbm_peak_per_gpu": round(out["value"] * 1e6 * ref_bytes / 1e9 / max(1, world) / HBM_PEAK_GBS, 4)}

    if rank == 0 and world == 1 and not args.no_extras:
        img = full.cpu().numpy()
        final_avg = img / np.float32(args.steps)
        ps = {}
        r.free()
        # prefixes of the same sample sequence (iterations 1..n), outside the timed region
        for n in (16, 64, 256, 1000):
            if n < args.steps:
                rr = capi.Renderer(scene, device=local_rank)
                rr.render(1, n)
                ps[f"{n}spp_vs_{args.steps}spp"] = round(psnr(rr.readback() / np.float32(n), final_avg), 2)
                rr.free()
        # cross-implementation check at the full sample count on whole rows
        from oracle import binding as ob
        rows = [540, 800] if args.steps >= 1000 else [100, 540, 800, 1079]
        threads = min(16, os.cpu_count() or 1)
        exact = True
        a_rows, l_rows = [], []
        for row in rows:
            ob.set_math_mode(ob.PORTABLE)
            ob.load_scene(scene_path, res=(W, H))
            ref = ob.render(1, args.steps, depth=DEPTH, variant=ob.RETIRE, nthreads=threads, pix_begin=row * W, pix_count=W)
            got = img[row * W:(row + 1) * W]
            exact = exact and bool(np.array_equal(got.view(np.uint32), ref.view(np.uint32)))
            ob.set_math_mode(ob.LIBM)
            ref_l = ob.render(1, args.steps, depth=DEPTH, variant=ob.LITERAL, nthreads=threads, pix_begin=row * W, pix_count=W)
            a_rows.append(got / np.float32(args.steps))
            l_rows.append(ref_l / np.float32(args.steps))
        ps["rows_checked"] = rows
        ps["gpu_bit_exact_vs_oracle_portable"] = exact
        ps["gpu_vs_reference_semantics_libm_db"] = round(psnr(np.concatenate(a_rows), np.concatenate(l_rows)), 2)
        out["psnr"] = ps
        out["cpu_baseline"] = cpu_baseline(scene_path)
        if args.save:
            capi.save_png(args.save + ".png", img, W, H, float(args.steps))
            capi.save_pfm(args.save + ".pfm", img, W, H, float(args.steps))
    else:
        r.free()

    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
