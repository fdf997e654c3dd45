#!/usr/bin/env python3
"""Fuzz the work distribution of the fused kernels (k_primary's strands in pieces, k_paths' pieces from counters, waves dealt by
measured work) against the UNFUSED stage kernels — an independent implementation of every depth with none of that machinery — on
random tiles, batch sizes, queue counts, grid sizes and piece knobs: exact mode, images compared bit for bit (GPU against GPU, so
moderate image sizes cost milliseconds).
usage: tools/fuzz_schedules.py [first_seed] [count]"""
import os, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cosc_4397_pathtracing_raytracing_project_amd import capi, scenes, parallel

first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
count = int(sys.argv[2]) if len(sys.argv) > 2 else 200
d = tempfile.mkdtemp()
bad = 0
for seed in range(first, first + count):
    rs = np.random.RandomState(seed)
    w, h = int(rs.randint(64, 720)), int(rs.randint(40, 400))
    kind = rs.randint(4)
    if kind == 0:
        text = scenes.cornell_scene_text(res=(w, h))
    elif kind == 1:
        text = scenes.stress_scene_text((6, 5, 4), res=(w, h))
    elif kind == 2:
        text = scenes.stress_scene_text((10, 10, 8), res=(w, h))
    else:
        text = scenes.random_scene_text(seed, int(rs.choice([12, 40, 150, 700])), res=(w, h), depth=int(rs.choice([2, 5, 8, 12])), clustered=bool(rs.randint(2)))
    path = scenes.write_scene(text, os.path.join(d, f"s{seed}.txt"))
    sc = capi.Scene(path, res=(w, h))
    world = int(rs.choice([1, 1, 2, 3, 5, 8]))
    rank = int(rs.randint(world))
    tile = parallel.striped_tile_for_rank(w, h, rank, world) if world > 1 else {}
    K = int(rs.choice([0, 1, 2, 3, 5, 8, 17, 40, 130]))
    spp = int(min(3 * max(K, 1) + 1, rs.choice([4, 9, 23, 60])))
    kw = dict(iters_per_batch=K, num_queues=int(rs.choice([0, 0, 4, 16, 64, 256, 1024])), blocks_per_cu=int(rs.choice([0, 0, 1, 2, 3])))
    env = {}
    if rs.randint(2): env["PT_PATHS_MIN_PIECE"] = str(int(rs.choice([1, 2, 8, 32])))
    if rs.randint(2): env["PT_PRIMARY_PIECES"] = str(int(rs.choice([2, 3, 5])))
    if rs.randint(4) == 0: env["PT_PATHS_PIECES"] = str(int(rs.choice([1, 3, 4])))
    for k in ("PT_PATHS_MIN_PIECE", "PT_PRIMARY_PIECES", "PT_PATHS_PIECES"):
        os.environ.pop(k, None)
    os.environ.update(env)
    imgs = []
    for unf in (False, True):
        r = capi.Renderer(sc, unfused_primary=unf, unfused_bounces=unf, **kw, **tile)
        r.render(1, spp)
        imgs.append(r.readback())
        pw = r.stats().paths_waves
        r.free()
        if not unf: dealt = pw
    ok = np.array_equal(imgs[0].view(np.uint32), imgs[1].view(np.uint32)) and np.isfinite(imgs[0]).all()
    bad += not ok
    print(f"seed {seed}: kind {kind} {w}x{h} tile {rank}/{world} spp {spp} {kw} {env} dealt {dealt >> 16}-{dealt & 0xffff}: {'ok' if ok else 'MISMATCH'}", flush=True)
print("mismatches:", bad)
sys.exit(1 if bad else 0)
