#!/usr/bin/env python3
"""Event counts of the grid walk per group of 64 rays (diagnostic build: tools/build_variant.sh walkstats "-DPT_WALK_STATS", then
PT_AMD_LIB=build/variants/walkstats.so tools/walk_stats.py [stress|random SEED N [clustered]] [spp]).  Counts cover k_primary and
every k_bounce_big launch of the run (fast arithmetic)."""
import ctypes as C, os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cosc_4397_pathtracing_raytracing_project_amd import capi, scenes

args = sys.argv[1:]
res = (1920, 1080)
if args and args[0] == "random":
    text = scenes.random_scene_text(int(args[1]), int(args[2]), res=res, clustered=len(args) > 3 and args[3] == "clustered")
    args = args[4 if len(args) > 3 and args[3] == "clustered" else 3:]
else:
    text = scenes.stress_scene_text(res=res)
    args = args[1:]
spp = int(args[0]) if args else 25
path = scenes.write_scene(text, os.path.join(tempfile.mkdtemp(), "s.txt"))
sc = capi.Scene(path, res=res)
L = capi.lib()
L.pt_debug_walk_stats.argtypes = [C.POINTER(C.c_ulonglong)]
buf = (C.c_ulonglong * 16)()
r = capi.Renderer(sc, arith="fast", debug_flags=256)
L.pt_debug_walk_stats(buf)  # drop what pt_init's probe counted
r.render(1, spp); r.sync()
L.pt_debug_walk_stats(buf)
st = r.stats()
g = buf[0]
names = ["groups", "steps", "walking lanes (sum over steps)", "filing rounds", "box-test chunks", "records tested", "primitive chunks",
         "candidates tested", "filing trips (max records of a lane, sum over rounds)", "filing lanes (sum over rounds)",
         "records skipped: handled in the cell the ray came from", "records passing (box test and closer-hit cull) = candidates filed",
         "records passing the box test"]
print(f"grid cells {st.grid_cells}, tight leaves {st.tight_leaves}, rays {sum(st.live_rays[:8])}, groups {g}")
for i, n in enumerate(names):
    print(f"  {n:60s} {buf[i]:14d}  per group {buf[i] / max(g, 1):8.2f}")
