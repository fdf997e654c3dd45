"""The oracle against every known answer we have for the reference (SURVEY.md §4 table,
tests/golden/survey_kats.json) and against the matrices its own utilities.cpp + GLM produce
(tests/golden/ref_xforms.json, made by oracle/ref_xform_harness.cpp).  CPU only."""
import ctypes as C
import json
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
KATS = json.load(open(os.path.join(HERE, "golden", "survey_kats.json")))
XF = json.load(open(os.path.join(HERE, "golden", "ref_xforms.json")))


def test_utilhash(oracle):
    for k, v in KATS["utilhash"].items():
        assert oracle.lib().orc_utilhash(int(k)) == v


def test_rng_seeds_and_draws(oracle):
    L = oracle.lib()
    for s in KATS["seeds"]:
        h = L.orc_seed(s["iter"], s["index"], s["depth"])
        assert h == s["h"]
        raw = (C.c_uint32 * 3)()
        u = (C.c_float * 3)()
        L.orc_rng_draws(h, 3, raw, u)
        assert list(raw)[:2] == s["raw"]
        # the survey printed 9 significant digits: exact float32 values round-trip through that
        assert [float(np.float32(x)) for x in u] == [float(np.float32(x)) for x in s["u01"]]
    assert L.orc_minstd_nth(1, 10000) == KATS["minstd_default_10000th"]


def test_u01_range(oracle):
    # u = float(x-1)/2^31 lies in [0,1] and can round to exactly 1.0
    L = oracle.lib()
    raw = (C.c_uint32 * 1)()
    u = (C.c_float * 1)()
    L.orc_rng_draws(1, 1, raw, u)
    assert raw[0] == 48271 and u[0] == np.float32(48270) / np.float32(2 ** 31)
    assert np.float32(2147483646 - 1) / np.float32(2 ** 31) == np.float32(1.0)


def test_glm_transforms_match_reference_build(oracle):
    """Bit-for-bit against matrices computed by the reference's own utilities.cpp + GLM."""
    for x in XF["xforms"]:
        trs = np.array(x["trs"], np.uint32).view(np.float32)
        m, i, it = oracle.build_xform(trs)
        assert np.array_equal(m.view(np.uint32), np.array(x["transform"], np.uint32))
        assert np.array_equal(i.view(np.uint32), np.array(x["inverse"], np.uint32))
        assert np.array_equal(it.view(np.uint32), np.array(x["invTranspose"], np.uint32))


def test_glm_vector_ops_match_reference_build(oracle):
    m, _, _ = oracle.build_xform(np.array([1.25, -2.5, 3.75, 30, 45, 60, 1, 2, 3], np.float32))
    for v in XF["vecops"]:
        a = np.array(v["a"], np.uint32).view(np.float32)
        b = np.array(v["b"], np.uint32).view(np.float32)
        exp = np.array(v["normalize_a"] + v["cross"] + [v["dot"], v["length_a"]] + v["M_point"] + v["M_dir"], np.uint32)
        assert np.array_equal(oracle.vecops(a, b, m).view(np.uint32), exp)


def test_camera_fixup(oracle, scene_dir):
    oracle.load_scene(scene_dir["cornell"])
    c = oracle.camera()
    k = KATS["camera_after_fixup"]
    for name in ("position", "view", "up", "right"):
        got = np.array(list(getattr(c, name)), np.float32)
        exp = np.array(k[name], np.float32)
        assert np.array_equal(got, exp), name
        assert np.array_equal(np.signbit(got), np.signbit(exp)), name  # -0 vs +0 matters downstream
    assert list(c.res) == [800, 800]
    assert oracle.trace_depth() == 8


def test_loader_quirks(oracle, tmp_path):
    """scene.cpp quirks: out-of-sequence ids are skipped; '//' lines ignored at top level;
    last line without newline still parsed; CRLF accepted (utilities.cpp:78-112)."""
    from cosc_4397_pathtracing_raytracing_project_amd import scenes
    txt = scenes.sphere_scene_text()
    # duplicate OBJECT 0 block with a wrong id (5): must be ignored
    txt2 = "// comment line\r\n" + txt.replace("\n", "\r\n") + "OBJECT 5\r\nsphere\r\nmaterial 0\r\nTRANS 1 1 1\r\n\r\nOBJECT 1\r\ncube\r\nmaterial 0\r\nSCALE 2 2 2"
    p = tmp_path / "quirk.txt"
    p.write_bytes(txt2.encode())
    oracle.load_scene(str(p))
    assert oracle.lib().orc_num_geoms() == 2
    g = oracle.geoms()
    assert g[0].type == 0 and g[1].type == 1
    assert g[1].transform[0] == 2.0 and g[1].transform[12] == 0.0  # SCALE parsed from the unterminated last line
    with pytest.raises(FileNotFoundError):
        oracle.load_scene(str(tmp_path / "missing.txt"))


def test_cornell_bvh_table(oracle, scene_dir):
    oracle.load_scene(scene_dir["cornell"])
    nodes = oracle.bvh()
    assert len(nodes) == len(KATS["cornell_bvh"]) == 13
    for n, (l, r, g, lo, hi) in zip(nodes, KATS["cornell_bvh"]):
        assert (n.left, n.right, n.geomIndex) == (l, r, g)
        # the survey printed 6 significant digits
        assert np.allclose(list(n.bmin), lo, rtol=2e-6, atol=1e-12)
        assert np.allclose(list(n.bmax), hi, rtol=2e-6, atol=1e-12)


def test_center_ray_hits(oracle, scene_dir):
    oracle.load_scene(scene_dir["cornell"])
    c = oracle.camera()
    o = np.array(list(c.position), np.float32)
    d = np.array(list(c.view), np.float32)
    fp = C.POINTER(C.c_float)
    pt = (C.c_float * 3)()
    nr = (C.c_float * 3)()
    out = C.c_int()
    res = {}
    for g in range(7):
        t = oracle.lib().orc_geom_test(g, o.ctypes.data_as(fp), d.ctypes.data_as(fp), pt, nr, C.byref(out))
        res[g] = (np.float32(t), np.array(list(pt), np.float32), np.array(list(nr), np.float32), out.value)
    k = KATS["cornell_center_ray"]
    for g in k["misses"]:
        assert res[g][0] == np.float32(-1.0)
    for g, key in ((3, "geom3"), (6, "geom6")):
        t, p, n, outside = res[g]
        assert t == np.float32(k[key]["t"])
        assert np.allclose(p, k[key]["p"], rtol=3e-6, atol=1e-15)
        assert np.allclose(n, k[key]["n"], rtol=3e-6, atol=1e-9)
        assert outside == k[key]["outside"]


@pytest.mark.parametrize("case", [c for c in KATS["images"] if c["res"][0] * c["res"][1] * c["spp"] <= 256 * 256 * 16],
                         ids=lambda c: f"{c['scene']}-{c['res'][0]}x{c['res'][1]}-{c['spp']}spp")
def test_image_kats_small(oracle, scene_dir, case):
    _check_image_case(oracle, scene_dir, case)


def test_image_kat_800(oracle, scene_dir):
    _check_image_case(oracle, scene_dir, KATS["images"][2])


def _check_image_case(oracle, scene_dir, case):
    """Whole-pipeline known answers: mean RGB and one pixel of the averaged image, LITERAL
    loop, LIBM arithmetic (what the survey ran).  9 printed digits → compare as float32."""
    oracle.load_scene(scene_dir[case["scene"]], res=tuple(case["res"]))
    spp, depth = case["spp"], case["depth"]
    img = oracle.render(1, spp, depth=depth, variant=oracle.LITERAL, nthreads=8)
    avg = img / np.float32(spp)
    mean = avg.mean(axis=0, dtype=np.float64)
    assert np.allclose(mean, case["mean_rgb"], rtol=2e-7, atol=0), (mean, case["mean_rgb"])
    if "pixel" in case:
        assert np.array_equal(avg[case["pixel"]], np.array(case["pixel_rgb"], np.float32))
    if "border_blue" in case:
        assert avg[0, 2] == np.float32(case["border_blue"])  # 0.5^8: a primary miss multiplies sky 8 times
    if "hit_fraction" in case:
        o, d = oracle.generate(0, avg.shape[0])
        h = oracle.intersect(o, d)
        assert abs((h["t"] >= 0).mean() - case["hit_fraction"]) < 5e-5


def test_alive_fractions_and_stack(oracle, scene_dir):
    """Live-ray fractions per depth at 800x800 from the literal loop's counters."""
    oracle.load_scene(scene_dir["cornell"], res=(200, 200))
    _, st = oracle.render(1, 4, depth=8, variant=oracle.LITERAL, want_stats=True)
    n = 200 * 200 * 4
    frac = np.array(st["live"][:8]) / n
    assert np.allclose(frac, KATS["alive_fraction_800"], atol=0.012)
    assert st["max_stack"] == KATS["cornell_traversal"]["max_stack"]
