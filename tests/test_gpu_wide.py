"""The 4-wide surface-area tree of our own (k_paths mode 3, csrc/pt_wide.inc, pt_api.cpp build_wide):
forced (debug_flags 1024) against the oracle bit for bit and against the BVH scan (4096) on scenes from 7 to a few thousand
primitives, unusual layouts included; the fma / fast images do not depend on the structure either."""
import os

import numpy as np
import pytest

from cosc_4397_pathtracing_raytracing_project_amd import capi, scenes
from test_gpu_grid import bits, layouts, scene_text

pytestmark = pytest.mark.gpu


def render(sc, spp, **kw):
    r = capi.Renderer(sc, **kw)
    try:
        r.render(1, spp)
        return r.readback(), r.stats()
    finally:
        r.free()


@pytest.mark.parametrize("prims,clustered", [(7, False), (16, False), (33, False), (64, False), (156, False), (156, True), (500, False), (1500, True)])
def test_wide_tree_equals_oracle_on_random_scenes(oracle, tmp_path, prims, clustered):
    res, spp = (128, 96), 5
    text = scenes.cornell_scene_text(res=res) if prims == 7 else scenes.random_scene_text(300 + prims, prims - 6, res=res, clustered=clustered)
    path = scenes.write_scene(text, str(tmp_path / "s.txt"))
    sc = capi.Scene(path, res=res)
    wide, st = render(sc, spp, debug_flags=1024 | 512)
    assert st.wide_nodes > 0 and st.grid_cells == 0
    scan, st2 = render(sc, spp, debug_flags=4096 | 512)
    assert st2.wide_nodes == 0
    oracle.set_math_mode(oracle.PORTABLE)
    oracle.load_scene(path, res=res)
    ref = oracle.render(1, spp, depth=8, variant=oracle.RETIRE, nthreads=min(16, os.cpu_count() or 1))
    diff = (bits(wide) != bits(ref)).any(axis=1)
    assert not diff.any(), f"{prims} primitives ({st.wide_nodes} wide nodes): {diff.sum()} pixels differ from the oracle, first {np.flatnonzero(diff)[:8]}"
    assert np.array_equal(bits(wide), bits(scan))


@pytest.mark.parametrize("name", ["flat", "far", "lattice_axis_rays", "inside", "mixed_sizes", "offset"])
def test_wide_tree_on_unusual_layouts(oracle, tmp_path, name):
    objects, eye, lookat = layouts()[name]
    res, spp = (128, 96), 4
    path = scenes.write_scene(scene_text(objects, res, eye, lookat), str(tmp_path / f"{name}.txt"))
    sc = capi.Scene(path, res=res)
    wide, st = render(sc, spp, debug_flags=1024 | 512)
    assert st.wide_nodes > 0
    oracle.set_math_mode(oracle.PORTABLE)
    oracle.load_scene(path, res=res)
    ref = oracle.render(1, spp, depth=8, variant=oracle.RETIRE, nthreads=min(16, os.cpu_count() or 1))
    diff = (bits(wide) != bits(ref)).any(axis=1)
    assert not diff.any(), f"{name}: {diff.sum()} pixels differ from the oracle, first {np.flatnonzero(diff)[:8]}"


@pytest.mark.parametrize("arith", ["fma", "fast"])
def test_wide_tree_and_scan_give_the_same_image_in_every_mode(tmp_path, arith):
    res, spp = (160, 90), 6
    for text in (scenes.random_scene_text(77, 150, res=res), scenes.stress_scene_text((10, 10, 8), res=res)):
        path = scenes.write_scene(text, str(tmp_path / "s.txt"))
        sc = capi.Scene(path, res=res)
        a, st = render(sc, spp, debug_flags=1024 | 512, arith=arith)
        b, _ = render(sc, spp, debug_flags=4096 | 512, arith=arith)
        assert st.wide_nodes > 0
        assert np.array_equal(bits(a), bits(b))


def test_wide_tree_depths_and_small_tiles(oracle, tmp_path):
    """Trace depths 2 and 20, several batches, a tile of a few hundred pixels (slices that touch many sub-lists)."""
    res = (64, 40)
    path = scenes.write_scene(scenes.random_scene_text(901, 90, res=res), str(tmp_path / "s.txt"))
    for depth, spp, kw in ((2, 4, {}), (20, 3, {}), (8, 7, dict(iters_per_batch=2)), (8, 4, dict(pixel_begin=res[0] * 7 + 3, pixel_count=333))):
        sc = capi.Scene(path, res=res)
        sc.trace_depth = depth
        img, st = render(sc, spp, debug_flags=1024 | 512, **kw)
        assert st.wide_nodes > 0
        oracle.set_math_mode(oracle.PORTABLE)
        oracle.load_scene(path, res=res)
        ref = oracle.render(1, spp, depth=depth, variant=oracle.RETIRE, nthreads=min(16, os.cpu_count() or 1))
        if "pixel_begin" in kw:
            ref = ref[kw["pixel_begin"]:kw["pixel_begin"] + kw["pixel_count"]]
        assert np.array_equal(bits(img), bits(ref)), (depth, spp, kw)
