"""The C++ host side end to end on the GPU: the headless driver `pt_render` goes through the
pathtrace.h-compatible shim (pathtraceInit / pathtrace per iteration / pathtraceFree, queued iterations,
state.image refresh) and writes PNG + PFM; its float image must equal the C-ABI image bit for bit."""
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "cosc_4397_pathtracing_raytracing_project_amd", "pt_render")


def read_pfm(path):
    raw = open(path, "rb").read()
    head, dims, scale, body = raw.split(b"\n", 3)
    w, h = map(int, dims.split())
    assert head == b"PF" and float(scale) < 0
    return np.frombuffer(body, np.float32).reshape(h, w, 3)[::-1].reshape(-1, 3), w, h


def test_pt_render_matches_c_abi(scene_dir, tmp_path):
    from cosc_4397_pathtracing_raytracing_project_amd import capi
    assert os.path.exists(BIN), "pt_render not built"
    out = str(tmp_path / "img")
    res, spp = (160, 120), 70  # 70 iterations: the shim flushes its queue every 64 → two pt_render calls
    p = subprocess.run([BIN, scene_dir["cornell"], "--res", f"{res[0]}x{res[1]}", "--spp", str(spp), "--out", out, "--pfm", "--hdr"],
                       capture_output=True, text=True, timeout=120)
    assert p.returncode == 0, p.stderr
    assert "Msamples/s" in p.stdout and os.path.exists(f"{out}.{spp}samp.png")
    avg, w, h = read_pfm(f"{out}.{spp}samp.pfm")
    assert (w, h) == res
    # --hdr: the Radiance file of image::saveHDR for the same image (the writer itself is pinned on the CPU side)
    capi.save_hdr(str(tmp_path / "expect.hdr"), np.ascontiguousarray(avg), w, h, 1.0)
    assert open(f"{out}.{spp}samp.hdr", "rb").read() == open(tmp_path / "expect.hdr", "rb").read()
    sc = capi.Scene(scene_dir["cornell"], res=res)
    r = capi.Renderer(sc)
    try:
        r.render(1, spp)
        img = r.readback()
    finally:
        r.free()
    assert np.array_equal((img / np.float32(spp)).view(np.uint32), np.ascontiguousarray(avg).view(np.uint32))


def test_pt_render_usage_and_missing_file(tmp_path):
    p = subprocess.run([BIN], capture_output=True, text=True)
    assert p.returncode == 1 and "Usage" in p.stdout
    p = subprocess.run([BIN, str(tmp_path / "nope.txt")], capture_output=True, text=True)
    assert p.returncode == 1 and "cannot read scene file" in p.stderr


GLUE = os.path.join(ROOT, "oracle", "_ref", "ref_glue_demo")


@pytest.mark.skipif(not os.path.exists(GLUE), reason="oracle/_ref/ref_glue_demo not built (needs the reference mounted at build time)")
@pytest.mark.parametrize("scene,args,arith", [("cornell", ["5", "96", "64"], "exact"), ("ref_twisted", ["3"], "exact"),
                                              ("ref_quirks", [], "exact"), ("cornell", ["4", "160", "90"], "fast")])
def test_reference_side_glue_end_to_end(scene_dir, tmp_path, scene, args, arith):
    """INTEGRATION.md §2 for real: the glue of tests/integration/pathtrace_amd_glue.cpp, compiled against the
    reference's OWN headers and linked with the reference's OWN scene loader (scene.cpp + utilities.cpp, built in place
    into oracle/_ref/), drives libpt_amd.so the way main.cpp drives pathtrace.cu: Free, Init, pathtrace per iteration,
    Free.  The SUM image it leaves in scene->state.image equals what our own loader + C ABI render, bit for bit."""
    from cosc_4397_pathtracing_raytracing_project_amd import capi
    path = scene_dir[scene] if scene in scene_dir else os.path.join(ROOT, "tests", "golden", "scenes", scene + ".txt")
    out = str(tmp_path / "o.f32")
    env = dict(os.environ, PT_GLUE_ARITH=arith)
    p = subprocess.run([GLUE, path, out] + args, capture_output=True, text=True, timeout=300, env=env)
    assert p.returncode == 0, p.stderr[-2000:]
    res = (int(args[1]), int(args[2])) if len(args) >= 3 else None
    sc = capi.Scene(path, res=res)
    spp = int(args[0]) if args else sc.iterations
    r = capi.Renderer(sc, arith=arith)
    try:
        r.render(1, spp)
        img = r.readback()
    finally:
        r.free()
    got = np.fromfile(out, np.float32).reshape(-1, 3)
    assert got.shape == img.shape and np.array_equal(got.view(np.uint32), img.view(np.uint32))
    assert f"traced depth {sc.trace_depth}" in p.stderr
