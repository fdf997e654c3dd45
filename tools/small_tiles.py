#!/usr/bin/env python3
"""Per-GPU throughput of the tile one rank of N renders (row-interleaved tile of rank 0, cornell 1080p, fast), on ONE GPU: what
the multi-GPU partition costs a device before any exchange — small tiles mean ~25-path sub-lists, ~190 iterations per batch and
queues of a few chunks.  usage: tools/small_tiles.py [worlds, e.g. 1,2,4,8] [iterations per batch, 0 = auto, e.g. 0,25,50]"""
import sys, time, os, tempfile
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import numpy as np
from cosc_4397_pathtracing_raytracing_project_amd import capi, scenes, parallel
W,H=1920,1080
path=scenes.write_scene(scenes.cornell_scene_text(res=(W,H)), os.path.join(tempfile.mkdtemp(),'c.txt'))
sc=capi.Scene(path,res=(W,H))
worlds=[int(x) for x in (sys.argv[1] if len(sys.argv)>1 else "4,8").split(',')]
ks=[int(x) for x in (sys.argv[2] if len(sys.argv)>2 else "0,25,50,100").split(',')]
for world in worlds:
    topt=parallel.striped_tile_for_rank(W,H,0,world) if world>1 else dict(pixel_begin=0,pixel_count=W*H)
    n=topt['pixel_count']
    for K in ks:
        r=capi.Renderer(sc, arith='fast', iters_per_batch=K, time_kernels=True, **topt)
        r.render(1,200); r.sync()
        line=f"world {world} tile {n} px K={r.stats().iters_per_batch:3d}:"
        for steps in (20, 2000):
            best=1e9
            r.reset_stats()  # the k_paths average below: launches of the last leg only (whole batches)
            for rep in range(4):
                r.clear(); r.sync()
                t0=time.perf_counter(); r.render(1,steps); r.sync(); dt=time.perf_counter()-t0
                best=min(best,dt)
            line+=f"  {steps} steps {best*1e3:8.3f} ms = {n*steps/best/1e6:6.0f} M/s"
        st=r.stats()
        line+=f"   k_paths {st.intersect_ms/max(1,st.intersect_launches)*1e3:7.1f} us/launch, {st.grid_blocks} blocks, waves per queue {st.paths_waves>>16}-{st.paths_waves&0xffff}"
        print(line, flush=True)
        r.free()
