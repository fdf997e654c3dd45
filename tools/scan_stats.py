#!/usr/bin/env python3
"""Event counts of the BVH-scan search (k_paths mode 1 / k_primary's global-table form) per round of 64 lanes on the ladder's random
scenes (diagnostic build: tools/build_variant.sh walkstats "-DPT_WALK_STATS", then PT_AMD_LIB=build/variants/walkstats.so
tools/scan_stats.py [--sizes 64,156,500] [--spp 25] [--flags 512]).  Fast arithmetic; the counters cover k_primary and k_paths."""
import argparse, ctypes as C, os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cosc_4397_pathtracing_raytracing_project_amd import capi, scenes
ap = argparse.ArgumentParser()
ap.add_argument("--sizes", default="64,156,500")
ap.add_argument("--spp", type=int, default=25)
ap.add_argument("--flags", type=int, default=512)
a = ap.parse_args()
res = (1920, 1080)
L = capi.lib()
L.pt_debug_walk_stats.argtypes = [C.POINTER(C.c_ulonglong)]
buf = (C.c_ulonglong * 16)()
d = tempfile.mkdtemp()
for size in a.sizes.split(","):
    prims = int(size)
    text = scenes.random_scene_text(100 + prims, prims - 6, res=res)
    sc = capi.Scene(scenes.write_scene(text, os.path.join(d, f"s{size}.txt")), res=res)
    r = capi.Renderer(sc, arith="fast", debug_flags=a.flags)
    L.pt_debug_walk_stats(buf)
    r.render(1, a.spp); r.sync()
    L.pt_debug_walk_stats(buf)
    st = r.stats(); r.free()
    rays = sum(st.live_rays[:8])
    g, gp = buf[0], buf[14]
    print(f"{prims} primitives, {len(sc.bvh())} nodes, top {st.num_top if hasattr(st, 'num_top') else '?'}: rays {rays}, searches {g} (k_paths rounds {gp}), "
          f"lanes searched per round: all {buf[13] / max(g, 1):.1f}, k_paths {buf[15] / max(gp, 1):.1f}")
    print(f"   per search: top entries passed {buf[8] / max(g, 1):.1f} ({buf[8] / max(buf[13], 1):.2f} per ray), scan steps {buf[1] / max(g, 1):.1f}, "
          f"active lanes per step {buf[2] / max(buf[1], 1):.1f}, node visits per ray {buf[2] / max(buf[13], 1):.1f}, steal steps {buf[3] / max(g, 1):.2f}, "
          f"chunks {buf[6] / max(g, 1):.2f}, candidates {buf[7] / max(g, 1):.1f} ({buf[7] / max(buf[13], 1):.2f} per ray)", flush=True)
