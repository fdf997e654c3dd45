"""Arithmetic modes of the kernels (PtOptions.arith, include/pt_amd.h).

`exact` is the parity anchor: bit-identical to the oracle, covered by every other GPU test.  `fma` (FMA contraction,
what nvcc does to the reference's kernels by default) and `fast` (fma + hardware rcp / rsq / sqrt / sin / cos,
float-only direction sampling) are held to the tolerance SURVEY.md §8(c) states for "same scene, same seed" against the
REFERENCE semantics — the oracle in LIBM mode running the reference-literal loop (pinned by the survey KATs):
  (1) every value finite;
  (2) at <= 16 spp at least 99.8 % of the pixels within 1e-5 (absolute, averaged radiance) — the rest are hit/miss
      flips at silhouettes, which the survey measured at 0.08 % between two compilations of the reference itself;
  (3) PSNR >= 45 dB + 10 log10(spp / 8).
Bound (2) was calibrated by the survey on the reference's own scenes.  Scenes with many small spheres are chaotic (a
sphere bounce multiplies a direction difference by distance / radius), so there even the 1-ulp difference between glibc's
and the portable sinf / cosf — two equally valid evaluations of the reference's source — changes more than 0.2 % of the
pixels (stress_big 160x90 at 8 spp: 0.39 %).  For those scenes the bound is therefore 0.2 % + 3 x that noise floor,
the floor being measured in the test itself (oracle PORTABLE vs oracle LIBM, CPU only); (1) and (3) stay absolute.

Primary rays are traced with the reference's exact arithmetic in every mode (pt_kernels.hip, namespace ex): they are the
same in every iteration of a pixel, so a tie broken differently there (the diagonals of a square cornell frame look
exactly along the box's corner edges) would not average out.
"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
MODES = ("fma", "fast")


def psnr(a, b):
    mse = np.mean((a.astype(np.float64) - b.astype(np.float64)) ** 2)
    return float("inf") if mse == 0 else 10.0 * np.log10(1.0 / mse)


def gpu(scene_path, res, spp, arith, depth=8, **kw):
    from cosc_4397_pathtracing_raytracing_project_amd import capi
    sc = capi.Scene(scene_path, res=res)
    sc.trace_depth = depth
    r = capi.Renderer(sc, arith=arith, **kw)
    try:
        r.render(1, spp)
        img = r.readback()
        st = r.stats()
    finally:
        r.free()
    assert st.arith == capi.ARITH[arith]
    return img


def check_tolerance(img, ref, spp, what, floor=0.0):
    """floor: fraction of pixels by which two valid evaluations of the reference (oracle PORTABLE vs LIBM) differ."""
    a, b = img / np.float32(spp), ref / np.float32(spp)
    assert np.isfinite(a).all(), what
    off = float((np.abs(a - b).max(axis=1) > 1e-5).mean())
    db = psnr(a, b)
    print(f"{what}: {100 * off:.3f} % of pixels off by > 1e-5 (noise floor {100 * floor:.3f} %), PSNR {db:.1f} dB")
    assert off <= 0.002 + 3 * floor, (what, off, floor)
    assert db >= 45.0 + 10 * np.log10(spp / 8), (what, db)
    return off, db


def noise_floor(oracle, ref_libm, spp, **render_kw):
    """Pixels that differ between the oracle's two math modes (same loop): what 1 ulp of sinf / cosf alone changes."""
    oracle.set_math_mode(oracle.PORTABLE)
    alt = oracle.render(1, spp, variant=oracle.LITERAL, **render_kw)
    oracle.set_math_mode(oracle.LIBM)
    return float((np.abs(alt / np.float32(spp) - ref_libm / np.float32(spp)).max(axis=1) > 1e-5).mean())


@pytest.mark.parametrize("arith", MODES)
@pytest.mark.parametrize("scene,res,spp,depth", [
    ("cornell", (256, 256), 16, 8),
    ("cornell", (800, 800), 8, 8),     # the configuration of the survey's FMA on/off experiment (51.2 dB, 0.08 %)
    ("sphere", (256, 256), 16, 4),     # BASELINE config C1
    ("stress", (160, 90), 8, 8),
    ("stress_big", (160, 90), 8, 8),   # global-memory tables, subtree scans
])
def test_mode_within_stated_tolerance_of_reference_semantics(scene_dir, oracle, arith, scene, res, spp, depth):
    img = gpu(scene_dir[scene], res, spp, arith, depth)
    oracle.set_math_mode(oracle.LIBM)
    oracle.load_scene(scene_dir[scene], res=res)
    ref = oracle.render(1, spp, depth=depth, variant=oracle.LITERAL, nthreads=16)
    floor = noise_floor(oracle, ref, spp, depth=depth, nthreads=16) if scene.startswith("stress") else 0.0
    check_tolerance(img, ref, spp, f"{arith} {scene} {res} {spp} spp", floor)


@pytest.mark.parametrize("arith", MODES)
def test_mode_c5_rows_within_tolerance(oracle, tmp_path, arith):
    """BASELINE config C5 at full size (10,170 primitives, 1080p): three rows against the reference semantics."""
    from cosc_4397_pathtracing_raytracing_project_amd import scenes
    w, h, spp = 1920, 1080, 8
    path = scenes.write_scene(scenes.stress_scene_text((22, 22, 21), res=(w, h), depth=8), str(tmp_path / "c5.txt"))
    img = gpu(path, (w, h), spp, arith)
    assert np.isfinite(img).all()
    oracle.set_math_mode(oracle.LIBM)
    oracle.load_scene(path, res=(w, h))
    rows = (300, 540, 900)
    nt = min(16, os.cpu_count() or 1)
    ref = np.concatenate([oracle.render(1, spp, depth=8, variant=oracle.LITERAL, nthreads=nt, pix_begin=r * w, pix_count=w) for r in rows])
    oracle.set_math_mode(oracle.PORTABLE)
    alt = np.concatenate([oracle.render(1, spp, depth=8, variant=oracle.LITERAL, nthreads=nt, pix_begin=r * w, pix_count=w) for r in rows])
    floor = float((np.abs(alt / np.float32(spp) - ref / np.float32(spp)).max(axis=1) > 1e-5).mean())
    got = np.concatenate([img[r * w:(r + 1) * w] for r in rows])
    check_tolerance(got, ref, spp, f"{arith} C5 rows {rows}", floor)


@pytest.mark.parametrize("arith", MODES)
def test_mode_random_scenes_within_tolerance(oracle, tmp_path, arith):
    """Rotated / non-uniformly scaled cubes and spheres, mirrors and rough mirrors (the specular lobe), emitters."""
    from cosc_4397_pathtracing_raytracing_project_amd import scenes
    for seed, n, clustered in ((2, 27, False), (3, 70, True), (4, 300, False)):
        res, spp = (128, 80), 8
        path = scenes.write_scene(scenes.random_scene_text(seed, n, res=res, clustered=clustered), str(tmp_path / f"r{seed}.txt"))
        img = gpu(path, res, spp, arith)
        oracle.set_math_mode(oracle.LIBM)
        oracle.load_scene(path, res=res)
        ref = oracle.render(1, spp, depth=8, variant=oracle.LITERAL, nthreads=16)
        floor = noise_floor(oracle, ref, spp, depth=8, nthreads=16)
        check_tolerance(img, ref, spp, f"{arith} random scene {seed}", floor)


@pytest.mark.parametrize("arith", MODES)
def test_mode_is_deterministic_and_layout_neutral(scene_dir, arith):
    """Within a mode the image is a pure function of (scene, iterations): batch size, queue count, tiles and the unfused
    kernel forms do not change a bit."""
    res, spp = (200, 120), 7
    a = gpu(scene_dir["cornell"], res, spp, arith)
    for kw in (dict(iters_per_batch=3), dict(num_queues=64, blocks_per_cu=2), dict(unfused_bounces=True), dict(unfused_primary=True)):
        b = gpu(scene_dir["cornell"], res, spp, arith, **kw)
        same = np.array_equal(a.view(np.uint32), b.view(np.uint32))
        if kw.get("unfused_primary"):
            # the fused primary kernel tests camera-relative boxes ((b - cam) * inv); the unfused one the general form:
            # identical in exact mode by construction, within tolerance otherwise
            assert same or psnr(a / spp, b / spp) > 60, kw
        else:
            assert same, kw
    n = res[0] * res[1]
    parts = [gpu(scene_dir["cornell"], res, spp, arith, pixel_begin=b0, pixel_count=c) for b0, c in ((0, 200 * 50), (200 * 50, n - 200 * 50))]
    assert np.array_equal(np.concatenate(parts).view(np.uint32), a.view(np.uint32))


def test_unknown_mode_and_release_build_reject_ablations(scene_dir):
    from cosc_4397_pathtracing_raytracing_project_amd import capi
    sc = capi.Scene(scene_dir["cornell"], res=(32, 32))
    with pytest.raises(capi.PtError):
        capi.Renderer(sc, arith=7)
    if not capi.lib().pt_library_has_ablations():
        for bit in (1, 2, 4, 8):
            with pytest.raises(capi.PtError, match="PT_ABLATE"):
                capi.Renderer(sc, debug_flags=bit)
    capi.pt_free()
