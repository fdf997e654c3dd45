#!/usr/bin/env python3
"""Scene-size ladder at 1080p: cornell.txt (7 primitives) and random scenes of 10 ... 1000 objects inside the cornell box
(+ its 6 walls), plus BASELINE config C5 (10,170 primitives) — Msamples/s per arithmetic mode with the library's own
choices, and with `--flags` extra debug_flags sets for A/B (e.g. 4096 = one launch per depth).  Images of all arms of
a scene are compared bit for bit within a mode.
usage: tools/scene_ladder.py [--spp N] [--arith fast,exact] [--flags 0,4096] [--sizes 7,16,26,58,150,494,994,c5]"""
import argparse, os, sys, tempfile, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cosc_4397_pathtracing_raytracing_project_amd import capi, scenes

ap = argparse.ArgumentParser()
ap.add_argument("--spp", type=int, default=100)
ap.add_argument("--arith", default="fast,exact")
ap.add_argument("--flags", default="0")
ap.add_argument("--sizes", default="7,16,32,64,156,500,1000,c5")
args = ap.parse_args()
res = (1920, 1080)
d = tempfile.mkdtemp()
for size in args.sizes.split(","):
    if size == "c5":
        text, prims = scenes.stress_scene_text(res=res), 10170
    elif int(size) == 7:
        text, prims = scenes.cornell_scene_text(res=res), 7
    else:
        prims = int(size)
        text = scenes.random_scene_text(100 + prims, prims - 6, res=res)  # + the six walls / light of the cornell box
    path = scenes.write_scene(text, os.path.join(d, f"s{size}.txt"))
    sc = capi.Scene(path, res=res)
    line = f"{prims:6d} primitives ({len(sc.bvh()):5d} nodes)"
    for arith in args.arith.split(","):
        imgs = []
        for flags in (int(f) for f in args.flags.split(",")):
            best = 0.0
            for rep in range(2):
                r = capi.Renderer(sc, arith=arith, debug_flags=flags)
                r.render(1, 8); r.sync()
                t0 = time.perf_counter(); r.render(9, args.spp); r.sync(); dt = time.perf_counter() - t0
                best = max(best, res[0] * res[1] * args.spp / dt / 1e6)
                img = r.readback(); st = r.stats(); r.free()
            imgs.append(img)
            line += f" | {arith} flags {flags}{' grid %d cells' % st.grid_cells if st.grid_cells else ''}: {best:7.0f}"
        if len(imgs) > 1:
            line += " eq" if all(np.array_equal(imgs[0].view(np.uint32), x.view(np.uint32)) for x in imgs[1:]) else " IMAGES DIFFER"
    print(line, flush=True)
