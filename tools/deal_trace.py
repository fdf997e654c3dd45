#!/usr/bin/env python3
"""How the dealing of k_paths' waves settles: fewest / most waves per queue and the batch time, batch by batch.
usage: tools/deal_trace.py [world=1] [batches=24]"""
import sys, time, os, tempfile
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
from cosc_4397_pathtracing_raytracing_project_amd import capi, scenes, parallel
W,H=1920,1080
world=int(sys.argv[1]) if len(sys.argv)>1 else 1
nb=int(sys.argv[2]) if len(sys.argv)>2 else 24
path=scenes.write_scene(scenes.cornell_scene_text(res=(W,H)), os.path.join(tempfile.mkdtemp(),'c.txt'))
sc=capi.Scene(path,res=(W,H))
topt=parallel.striped_tile_for_rank(W,H,0,world) if world>1 else {}
r=capi.Renderer(sc, arith='fast', time_kernels=True, **topt)
K=r.stats().iters_per_batch
r.render(1,K); r.sync()
out=[]
for b in range(nb):
    r.reset_stats(); t0=time.perf_counter(); r.render(1+(b+1)*K, K); r.sync(); dt=time.perf_counter()-t0
    st=r.stats()
    out.append(f"{st.paths_waves>>16}-{st.paths_waves&0xffff} {dt*1e3:.3f}ms k_paths {st.intersect_ms/max(1,st.intersect_launches)*1e3:.0f}us")
print(f"world {world} K={K}: "+" | ".join(out))
