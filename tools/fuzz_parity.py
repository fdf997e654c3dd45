#!/usr/bin/env python3
"""Fuzz the GPU renderer against the oracle on random scenes (bit-exact, PORTABLE math).
usage: tools/fuzz_parity.py [first_seed] [count] [debug_flags] [large]   (debug_flags 256: the uniform-grid walk forced on every
scene, also mesh scenes every fifth seed; `large`: scenes of 600-5000 objects at 320x200 — the sizes whose sphere leaves get the
tightened traversal boxes, with enough rays to graze silhouettes)"""
import os, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cosc_4397_pathtracing_raytracing_project_amd import capi, scenes
from oracle import binding as ob

first = int(sys.argv[1]) if len(sys.argv) > 1 else 100
count = int(sys.argv[2]) if len(sys.argv) > 2 else 40
flags = int(sys.argv[3]) if len(sys.argv) > 3 else 0
large = len(sys.argv) > 4 and sys.argv[4] == "large"
ob.build(); ob.set_math_mode(ob.PORTABLE)
bad = 0
d = tempfile.mkdtemp()
for seed in range(first, first + count):
    rs = np.random.RandomState(seed)
    n = int(rs.choice([2, 10, 40, 150, 600, 2500]))
    res = (int(rs.choice([64, 96, 130])), int(rs.choice([48, 64])))
    depth = int(rs.choice([1, 3, 8, 12]))
    spp = int(rs.choice([2, 5]))
    if large:
        n, res, spp = int(rs.choice([600, 1500, 2500, 5000])), (320, 200), int(rs.choice([2, 3]))
    clustered = bool(rs.randint(2))
    kw = [dict(), dict(unfused_bounces=True), dict(unfused_primary=True), dict(iters_per_batch=1), dict(num_queues=4)][rs.randint(5)]
    text = scenes.random_scene_text(seed, n, res=res, depth=depth, clustered=clustered)
    if flags and seed % 5 == 0:
        text = scenes.mesh_scene_text(res=res, grid=int(rs.choice([0, 3, 5])))
        depth = 8
    path = scenes.write_scene(text, os.path.join(d, f"s{seed}.txt"))
    sc = capi.Scene(path, res=res)
    kw = dict(kw, debug_flags=flags)
    # every other seed: k_paths' pieces of three paths and k_primary's strands in three pieces, so that small images go through the
    # piece switches, the guarded counter ring and the strand counter (same image)
    for name in ("PT_PATHS_MIN_PIECE", "PT_PRIMARY_PIECES"):
        if seed % 2:
            os.environ[name] = "3"
        else:
            os.environ.pop(name, None)
    r = capi.Renderer(sc, **kw); r.render(1, spp); img = r.readback(); r_tight = r.stats().tight_leaves; r.free()
    ob.load_scene(path, res=res)
    ref = ob.render(1, spp, depth=depth, variant=ob.RETIRE, nthreads=min(16, os.cpu_count() or 1))
    ok = np.array_equal(img.view(np.uint32), ref.view(np.uint32))
    bad += not ok
    print(f"seed {seed}: {n} objects {res} depth {depth} spp {spp} clustered {clustered} {kw} tight leaves {r_tight}: {'ok' if ok else 'MISMATCH'}", flush=True)
print("mismatches:", bad)
sys.exit(1 if bad else 0)
