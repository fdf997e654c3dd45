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
