#!/usr/bin/env python3
"""Event counts of the wide-tree walk (k_paths mode 3) per round of a wave on the ladder's random scenes (diagnostic build:
tools/build_variant.sh walkstats "-DPT_WALK_STATS", then PT_AMD_LIB=build/variants/walkstats.so tools/wide_stats.py
[--sizes 32,64,156,500] [--spp 25] [--flags 1536]).  Fast arithmetic."""
import argparse, ctypes as C, os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cosc_4397_pathtracing_raytracing_project_amd import capi, scenes
ap = argparse.ArgumentParser()
ap.add_argument("--sizes", default="32,64,156,500")
ap.add_argument("--spp", type=int, default=25)
ap.add_argument("--flags", type=int, default=1024 | 512)
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
    rays = sum(st.live_rays[1:8])
    rounds = max(buf[14], 1)
    print(f"{prims} primitives, {st.wide_nodes} wide nodes: bounce rays {rays}, rounds {buf[14]}, live lanes per round {buf[15] / rounds:.1f}, "
          f"shaded per round {buf[13] / rounds:.1f}")
    print(f"   per round: steps {buf[1] / rounds:.2f}, walking lanes per step {buf[2] / max(buf[1], 1):.1f}; per ray: node visits {buf[2] / max(rays, 1):.2f}, "
          f"candidates {buf[7] / max(rays, 1):.2f}; chunks per round {buf[6] / rounds:.2f} ({buf[7] / max(buf[6], 1):.1f} entries each, forced partial {buf[3] / rounds:.2f})", flush=True)
