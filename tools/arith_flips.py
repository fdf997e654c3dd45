#!/usr/bin/env python3
"""Flip-rate of the arithmetic modes on the GPU: for every library given (variants from tools/arith_bisect.sh, or the
product library) render cornell at 256x256x16 spp and 800x800x8 spp in the fast mode and count the pixels that differ
from the reference semantics (oracle, LIBM, reference-literal loop) by more than 1e-5, with the PSNR.  One child
process per library (PT_AMD_LIB is read at import)."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
CASES = [("cornell", (256, 256), 16), ("cornell", (800, 800), 8), ("cornell", (1920, 1080), 2), ("stress", (160, 90), 8), ("stress_big", (160, 90), 8), ("random4", (128, 80), 8), ("random3", (128, 80), 8)]

def scene_text(name, res):
    from cosc_4397_pathtracing_raytracing_project_amd import scenes
    if name == "cornell":
        return scenes.cornell_scene_text()
    if name == "random4":
        return scenes.random_scene_text(4, 300, res=res)
    if name == "random3":
        return scenes.random_scene_text(3, 70, res=res, clustered=True)
    return scenes.stress_scene_text((6, 5, 4) if name == "stress" else (10, 10, 8), res=res)

def child(mode):
    import tempfile
    import numpy as np
    from cosc_4397_pathtracing_raytracing_project_amd import capi, scenes
    out = {}
    with tempfile.TemporaryDirectory() as td:
        for name, res, spp in CASES:
            path = scenes.write_scene(scene_text(name, res), os.path.join(td, "c.txt"))
            sc = capi.Scene(path, res=res)
            r = capi.Renderer(sc, arith=mode)
            r.render(1, spp)
            img = r.readback()
            r.free()
            np.save(os.path.join(os.environ["PT_FLIPS_DIR"], f"{os.environ['PT_FLIPS_TAG']}_{name}_{res[0]}.npy"), img)

def main():
    import tempfile
    import numpy as np
    from oracle import binding as ob
    from cosc_4397_pathtracing_raytracing_project_amd import scenes
    libs = sys.argv[1:] or [os.path.join(ROOT, "cosc_4397_pathtracing_raytracing_project_amd", "libpt_amd.so")]
    td = tempfile.mkdtemp()
    refs = {}
    ob.set_math_mode(ob.LIBM)
    for name, res, spp in CASES:
        path = scenes.write_scene(scene_text(name, res), os.path.join(td, "c.txt"))
        ob.load_scene(path, res=res)
        refs[(name, res[0])] = ob.render(1, spp, depth=8, variant=ob.LITERAL, nthreads=16)
    runs = []
    for lib in libs:
        tag = os.path.basename(lib).replace(".so", "")
        modes = ["exact", "fma", "fast"] if tag in ("all", "libpt_amd") else ["fast"]
        for mode in modes:
            runs.append((lib, f"{tag}:{mode}", mode))
    for lib, tag, mode in runs:
        env = dict(os.environ, PT_AMD_LIB=os.path.abspath(lib), PT_FLIPS_DIR=td, PT_FLIPS_TAG=tag.replace(":", "_"))
        subprocess.check_call([sys.executable, __file__, "--child", mode], env=env)
        line = f"{tag:24s}"
        for name, res, spp in CASES:
            img = np.load(os.path.join(td, f"{tag.replace(':', '_')}_{name}_{res[0]}.npy"))
            a, b = img / np.float32(spp), refs[(name, res[0])] / np.float32(spp)
            bad = np.abs(a - b).max(axis=1) > 1e-5
            mse = float(np.mean((a.astype(np.float64) - b) ** 2))
            line += f" | {name[:7]} {res[0]}x{spp}: {100 * bad.mean():.3f}% {bad.sum()}px {10 * np.log10(1 / max(mse, 1e-30)):.1f}dB"
        print(line, flush=True)

if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--child":
        child(sys.argv[2])
    else:
        main()
