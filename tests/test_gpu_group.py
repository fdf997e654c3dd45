"""The C++ single-process multi-GPU path (pt_group_*, csrc/pt_group.cpp; BASELINE config 4) as far as the box allows:
with the devices that are visible (one on the test box, eight on the driver's node) the group image must equal the
single-context image bit for bit, the on-device PNG conversion must equal the host writer's bytes, and `pt_render
--gpus K` must produce the same files as the pathtrace.h shim path."""
import os
import subprocess
import zlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "cosc_4397_pathtracing_raytracing_project_amd", "pt_render")


def visible_devices():
    import torch
    return torch.cuda.device_count()


def single(scene_path, res, spp, **kw):
    from cosc_4397_pathtracing_raytracing_project_amd import capi
    sc = capi.Scene(scene_path, res=res)
    r = capi.Renderer(sc, **kw)
    try:
        r.render(1, spp)
        return r.readback(), r.save_u8(spp)
    finally:
        r.free()


def expected_u8(img_sum, w, h, spp):
    """saveImage (main.cpp:86-107) + savePNG (image.cpp:22-39) in numpy: x mirror, clamp, * 255, truncate."""
    avg = (img_sum.reshape(h, w, 3) / np.float32(spp))[:, ::-1]
    return (np.minimum(np.maximum(avg, np.float32(0)), np.float32(1)) * np.float32(255)).astype(np.uint8)


def png_pixels(path):
    raw = open(path, "rb").read()
    assert raw[:8] == b"\x89PNG\r\n\x1a\n"
    pos, idat, w, h = 8, b"", 0, 0
    while pos < len(raw):
        n = int.from_bytes(raw[pos:pos + 4], "big")
        typ = raw[pos + 4:pos + 8]
        if typ == b"IHDR":
            w, h = int.from_bytes(raw[pos + 8:pos + 12], "big"), int.from_bytes(raw[pos + 12:pos + 16], "big")
        if typ == b"IDAT":
            idat += raw[pos + 8:pos + 8 + n]
        pos += 12 + n
    rows = np.frombuffer(zlib.decompress(idat), np.uint8).reshape(h, 1 + 3 * w)
    assert (rows[:, 0] == 0).all()
    return rows[:, 1:].reshape(h, w, 3)


@pytest.mark.parametrize("arith", ["exact", "fast"])
def test_group_of_visible_devices_equals_single_context(scene_dir, arith):
    from cosc_4397_pathtracing_raytracing_project_amd import capi
    res, spp = (160, 101), 9  # odd row count: devices own different numbers of rows
    ref, ref8 = single(scene_dir["cornell"], res, spp, arith=arith)
    ndev = visible_devices()
    for k in sorted({1, ndev}):
        g = capi.Group(capi.Scene(scene_dir["cornell"], res=res), list(range(k)), arith=arith, iters_per_batch=4)
        try:
            g.render(1, 5)
            g.render(6, 4)
            img = g.gather()
            u8 = g.gather_u8(spp)
            assert g.stats(0).samples == ((res[1] + k - 1) // k) * res[0] * spp
        finally:
            g.free()
        assert np.array_equal(img.view(np.uint32), ref.view(np.uint32)), f"{k} device(s)"
        assert np.array_equal(u8, ref8), f"{k} device(s)"
        assert np.array_equal(u8, expected_u8(ref, res[0], res[1], spp))


@pytest.mark.parametrize("k", [2, 3])
@pytest.mark.parametrize("scene,res,spp,kw", [
    ("cornell", (160, 101), 9, dict(arith="exact")),            # odd row count: contexts own 51/50 resp. 34/34/33 rows
    ("cornell", (96, 64), 7, dict(arith="fast")),
    ("stress_big", (160, 90), 4, dict(arith="exact", debug_flags=256)),  # the uniform-grid kernels (forced)
    ("stress_big", (160, 91), 4, dict(arith="exact", debug_flags=512)),  # the BVH-scan kernels for big scenes
])
def test_group_of_contexts_on_one_device(scene_dir, k, scene, res, spp, kw):
    """K contexts on device 0 (device list {0, 0, ...}): everything the multi-GPU write-out does except RCCL itself runs —
    per-context streams, striped tiles with global pixel indices, the exchange into the root's receive buffer at
    recv_off, the strided row placement, the float / u8 / preview variants — and must give the single-context image
    bit for bit.  (RCCL cannot place two ranks on one device, so the exchange is PT_GROUP_TRANSPORT_COPY here.)"""
    from cosc_4397_pathtracing_raytracing_project_amd import capi
    ref, ref8 = single(scene_dir[scene], res, spp, **kw)
    sc1 = capi.Scene(scene_dir[scene], res=res)
    r = capi.Renderer(sc1, **kw)
    try:
        r.render(1, spp)
        want_prev = r.preview(spp)
    finally:
        r.free()
    g = capi.Group(capi.Scene(scene_dir[scene], res=res), [0] * k, iters_per_batch=3, **kw)
    try:
        assert g.transport == "copy"
        g.render(1, spp - 2)
        g.render(spp - 1, 2)
        img = g.gather()
        u8 = g.gather_u8(spp)
        prev = g.preview(spp)
        img2 = g.gather()  # a second write-out reuses the group's buffers
        rows = [(res[1] - i + k - 1) // k for i in range(k)]
        for i in range(k):
            assert g.stats(i).samples == rows[i] * res[0] * spp
    finally:
        g.free()
    assert np.array_equal(img.view(np.uint32), ref.view(np.uint32))
    assert np.array_equal(img2.view(np.uint32), ref.view(np.uint32))
    assert np.array_equal(u8, ref8) and np.array_equal(prev, want_prev)


def test_group_transport_selection(scene_dir):
    from cosc_4397_pathtracing_raytracing_project_amd import capi
    sc = capi.Scene(scene_dir["cornell"], res=(64, 48))
    with pytest.raises(capi.PtError, match="listed twice"):
        capi.Group(sc, [0, 0], transport="rccl")
    g = capi.Group(sc, [0], transport="auto")
    try:
        assert g.transport == "rccl"  # distinct devices: RCCL stays the default
    finally:
        g.free()
    ref, _ = single(scene_dir["cornell"], (64, 48), 3)
    g = capi.Group(capi.Scene(scene_dir["cornell"], res=(64, 48)), [0, 0, 0, 0, 0], transport="copy")
    try:
        g.render(1, 3)
        assert np.array_equal(g.gather().view(np.uint32), ref.view(np.uint32))
    finally:
        g.free()


def test_pt_render_device_list_on_one_gpu(scene_dir, tmp_path):
    """`pt_render --devices 0,0,0`: the C++ multi-device driver with three contexts on the one card."""
    outs = {}
    for tag, extra in (("shim", []), ("d3", ["--devices", "0,0,0"]), ("d2p", ["--devices", "0,0", "--preview", "4"])):
        out = str(tmp_path / tag)
        p = subprocess.run([BIN, scene_dir["cornell"], "--res", "160x121", "--spp", "10", "--out", out, "--pfm"] + extra,
                           capture_output=True, text=True, timeout=300)
        assert p.returncode == 0, p.stderr + p.stdout
        outs[tag] = (open(f"{out}.10samp.png", "rb").read(), open(f"{out}.10samp.pfm", "rb").read())
    assert outs["d3"] == outs["shim"] and outs["d2p"] == outs["shim"]
    assert os.path.exists(str(tmp_path / "d2p") + ".preview.png")
    p = subprocess.run([BIN, scene_dir["cornell"], "--devices", "0,0", "--transport", "rccl"], capture_output=True, text=True, timeout=60)
    assert p.returncode != 0 and "listed twice" in p.stderr


def test_group_progressive_preview(scene_dir, tmp_path):
    """pt_group_preview_rgba8 (sendImageToPBO on every device + one exchange) equals the single-context preview, can be
    taken between batches without disturbing the accumulation, and `pt_render --gpus K --preview N` writes the file."""
    from cosc_4397_pathtracing_raytracing_project_amd import capi
    res, spp = (96, 65), 6
    sc = capi.Scene(scene_dir["cornell"], res=res)
    r = capi.Renderer(sc)
    try:
        r.render(1, 4)
        want4 = r.preview(4)
        r.render(5, 2)
        want = r.readback()
    finally:
        r.free()
    g = capi.Group(capi.Scene(scene_dir["cornell"], res=res), list(range(visible_devices())))
    try:
        g.render(1, 4)
        got4 = g.preview(4)
        g.render(5, 2)
        img = g.gather()
    finally:
        g.free()
    assert np.array_equal(got4, want4) and np.array_equal(img.view(np.uint32), want.view(np.uint32))
    out = str(tmp_path / "p")
    p = subprocess.run([BIN, scene_dir["cornell"], "--res", "64x48", "--spp", "9", "--gpus", "1", "--preview", "4", "--out", out],
                       capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr
    assert os.path.exists(out + ".preview.png") and png_pixels(out + ".preview.png").shape == (48, 64, 3)
    q = subprocess.run([BIN, scene_dir["cornell"], "--res", "64x48", "--spp", "9", "--gpus", "1", "--out", out + "n"],
                       capture_output=True, text=True, timeout=300)
    assert q.returncode == 0 and open(out + ".9samp.png", "rb").read() == open(out + "n.9samp.png", "rb").read()


def test_save_u8_on_device_equals_host_png_writer(scene_dir, tmp_path):
    from cosc_4397_pathtracing_raytracing_project_amd import capi
    res, spp = (96, 64), 5
    img, u8 = single(scene_dir["cornell"], res, spp)
    path = str(tmp_path / "host.png")
    capi.save_png(path, img, res[0], res[1], float(spp))
    assert np.array_equal(png_pixels(path), u8)
    path2 = str(tmp_path / "dev.png")
    capi.write_png_rgb8(path2, u8)
    assert open(path, "rb").read() == open(path2, "rb").read()
    # tiles that are not whole rows cannot be mirrored on the device
    sc = capi.Scene(scene_dir["cornell"], res=res)
    r = capi.Renderer(sc, pixel_begin=10, pixel_count=200)
    try:
        r.render(1, 1)
        with pytest.raises(capi.PtError, match="whole image rows"):
            r.save_u8(1.0)
    finally:
        r.free()


def test_device_conversion_matches_reference_compiled_writer(scene_dir):
    """k_save_u8 against tests/golden/ref_image.json (the reference's own writer compiled in place): same bytes for
    negatives, exact 1, byte boundaries, > 1, infinities and NaN, in every arithmetic mode's build of the kernel."""
    import json
    from cosc_4397_pathtracing_raytracing_project_amd import capi
    g = json.load(open(os.path.join(ROOT, "tests", "golden", "ref_image.json")))
    w, h = g["width"], g["height"]
    img = np.array(g["sum_bits"], np.uint32).view(np.float32).reshape(h * w, 3)
    want = np.array(g["png_rgb8"], np.uint8).reshape(h, w, 3)
    for arith in ("exact", "fma", "fast"):
        r = capi.Renderer(capi.Scene(scene_dir["cornell"], res=(16, 16)), arith=arith)
        try:
            assert np.array_equal(capi.Renderer.stage_save_u8(img, w, h, float(g["samples"])), want), arith
        finally:
            r.free()


def test_pt_render_gpus_matches_shim_path(scene_dir, tmp_path):
    """`pt_render --gpus 1` (pt_group + RCCL communicator + device-side PNG bytes) against the default path
    (pathtrace.h shim + host PNG writer): identical PNG and PFM files; with every visible device too."""
    assert os.path.exists(BIN), "pt_render not built"
    res, spp = "160x120", "33"
    outs = {}
    ndev = visible_devices()
    runs = [("shim", []), ("g1", ["--gpus", "1"])] + ([("gall", ["--gpus", "0"])] if ndev > 1 else [])
    for tag, extra in runs:
        out = str(tmp_path / tag)
        p = subprocess.run([BIN, scene_dir["cornell"], "--res", res, "--spp", spp, "--out", out, "--pfm"] + extra,
                           capture_output=True, text=True, timeout=300)
        assert p.returncode == 0, p.stderr + p.stdout
        assert "Msamples/s" in p.stdout
        outs[tag] = (open(f"{out}.{spp}samp.png", "rb").read(), open(f"{out}.{spp}samp.pfm", "rb").read())
    for tag in outs:
        assert outs[tag] == outs["shim"], tag
    p = subprocess.run([BIN, scene_dir["cornell"], "--gpus", str(ndev + 1)], capture_output=True, text=True, timeout=60)
    assert p.returncode == 1 and "visible" in p.stderr


def test_pt_render_stamp_uses_reference_file_name(scene_dir, tmp_path):
    import re
    p = subprocess.run([BIN, scene_dir["cornell"], "--res", "32x32", "--spp", "2", "--stamp", "--arith", "fma"],
                       capture_output=True, text=True, timeout=120, cwd=str(tmp_path))
    assert p.returncode == 0, p.stderr
    names = os.listdir(tmp_path)
    # <FILE>.<UTC start time>.<samples>samp.png (main.cpp:99-102, preview.cpp:18-24)
    assert any(re.fullmatch(r"cornell\.\d{4}-\d\d-\d\d_\d\d-\d\d-\d\dz\.2samp\.png", n) for n in names), names


def test_bench_two_ranks_rehearsal_on_one_card(tmp_path):
    """The driver's multi-GPU launch line (`python -m torch.distributed.run --nproc-per-node N bench.py --gpus N`) with two
    ranks sharing the one card of the test box: RCCL needs one GPU per rank, so the tile gather goes through gloo
    (PT_DIST_BACKEND); everything else — striped tiles with global pixel indices, per-rank renderers, barrier, max-over-ranks
    timing, one JSON line from rank 0 — is the code path of the real run."""
    import json
    import sys
    env = dict(os.environ, PT_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", "29517", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "12", "--warmup", "3"],
                       capture_output=True, text=True, timeout=600, env=env, cwd=str(tmp_path))
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 12 and d["value"] > 0 and d["scaling"] == "strong"
    assert "interleaved-row tiles" in d["config"]["workload"] and d["config"]["arith"] == "fast"
    assert d["roofline"]["frac"] > 0 and "cpu_baseline" not in d


def test_gather_tiles_through_rccl_with_one_rank(tmp_path):
    """What a one-GPU box can run of the nccl (= RCCL) write-out: a one-rank process group on the card, the tile gather of
    parallel.gather_tiles through it (communicator creation, the collective call with device tensors, row placement of a
    striped tile).  More than one rank needs more than one GPU (RCCL refuses two ranks on one device): NOT RUN here."""
    import sys
    code = (
        "import os, sys, torch, torch.distributed as dist\n"
        f"sys.path.insert(0, {ROOT!r})\n"
        "from cosc_4397_pathtracing_raytracing_project_amd import parallel\n"
        "torch.cuda.set_device(0)\n"
        "dist.init_process_group('nccl', init_method='tcp://127.0.0.1:29519', world_size=1, rank=0, device_id=torch.device('cuda', 0))\n"
        "W, H = 64, 6\n"
        "tile = torch.arange(W * H * 3, dtype=torch.float32, device='cuda:0').view(W * H, 3)\n"
        "for striped in (False, True):\n"
        "    full = parallel.gather_tiles(tile, W, H, 0, 1, striped=striped, always_collective=True)\n"
        "    torch.cuda.synchronize()\n"
        "    assert full.is_cuda and torch.equal(full, tile), striped\n"
        "print('backend', dist.get_backend())\n"
        "dist.destroy_process_group()\n")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, env=env, cwd=str(tmp_path))
    assert p.returncode == 0, p.stderr[-2000:]
    assert "backend nccl" in p.stdout
