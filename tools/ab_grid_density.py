#!/usr/bin/env python3
"""Forced grid walk at several resolutions (PT_GRID_DENSITY = cells per primitive; unset = the host's search) on the random scenes
of tools/ab_random_scenes.py.  usage: tools/ab_grid_density.py [spp]"""
import os, sys, tempfile, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cosc_4397_pathtracing_raytracing_project_amd import capi, scenes

spp = int(sys.argv[1]) if len(sys.argv) > 1 else 200
res = (1920, 1080)
d = tempfile.mkdtemp()
for seed, n, clustered in [(6, 1500, True), (11, 5000, True), (8, 5000, False)]:
    path = scenes.write_scene(scenes.random_scene_text(seed, n, res=res, clustered=clustered), os.path.join(d, f"r{seed}.txt"))
    sc = capi.Scene(path, res=res)
    line = f"seed {seed}: {n} objects{' clustered' if clustered else ''}:"
    for dens in (None, 1.0, 2.0, 4.0, 8.0, 16.0):
        if dens is None:
            os.environ.pop("PT_GRID_DENSITY", None)
        else:
            os.environ["PT_GRID_DENSITY"] = str(dens)
        best = 0.0
        for rep in range(2):
            r = capi.Renderer(sc, arith="fast", debug_flags=256)
            r.render(1, 8); r.sync()
            t0 = time.perf_counter(); r.render(9, spp); r.sync(); dt = time.perf_counter() - t0
            best = max(best, res[0] * res[1] * spp / dt / 1e6)
            cells = r.stats().grid_cells; r.free()
        line += f" | {dens or 'search'}: {cells} cells {best:.0f}"
    print(line, flush=True)
