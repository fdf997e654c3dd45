#!/usr/bin/env python3
"""Grid walk against BVH scan (PtOptions.debug_flags 512) on random scenes — the check that the host's choice of the grid
(pt_api.cpp build_grid) does not hurt scenes that are not lattices.  usage: tools/ab_random_scenes.py [spp]"""
import os, sys, tempfile, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cosc_4397_pathtracing_raytracing_project_amd import capi, scenes

spp = int(sys.argv[1]) if len(sys.argv) > 1 else 200
res = (1920, 1080)
d = tempfile.mkdtemp()
cases = [(8, 5000, False), (9, 2000, False), (10, 20000, False), (6, 1500, True), (11, 5000, True)]
if len(sys.argv) > 2 and sys.argv[2] == "small":
    cases = [(12, 60, False), (13, 150, False), (4, 300, False), (14, 600, False), (15, 1000, False), (16, 300, True), ("lattice", 806, False)]
for seed, n, clustered in cases:
    text = scenes.stress_scene_text((10, 10, 8), res=res) if seed == "lattice" else scenes.random_scene_text(seed, n, res=res, clustered=clustered)
    path = scenes.write_scene(text, os.path.join(d, f"r{seed}.txt"))
    sc = capi.Scene(path, res=res)
    g = sc.grid(forced=True)
    line = f"seed {seed}: {n} objects{' clustered' if clustered else ''}, {len(sc.bvh())} nodes; grid " + \
           (f"{list(g[0].res)} {g[0].num_records / g[0].num_leaves:.1f} refs/leaf" if g else "not chosen")
    imgs = {}
    for flags in (0, 256, 512):  # the library's choice, grid forced, BVH scan forced
        best = 0.0
        for rep in range(3):
            r = capi.Renderer(sc, arith="fast", debug_flags=flags)
            r.render(1, 8); r.sync()
            t0 = time.perf_counter(); r.render(9, spp); r.sync(); dt = time.perf_counter() - t0
            best = max(best, res[0] * res[1] * spp / dt / 1e6)
            imgs[flags] = r.readback()
            st = r.stats(); r.free()
        line += f" | flags {flags} ({'grid' if st.grid_cells else 'bvh'}): {best:.0f} Msamples/s"
    a, b, c = list(imgs.values())
    line += f" | images equal: {np.array_equal(a.view(np.uint32), b.view(np.uint32)) and np.array_equal(a.view(np.uint32), c.view(np.uint32))}"
    print(line, flush=True)
