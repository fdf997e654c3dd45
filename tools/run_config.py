#!/usr/bin/env python3
"""Render one BASELINE configuration through the C ABI, report Msamples/s and check rows against the oracle.
usage: tools/run_config.py {cornell|sphere|stress} --res WxH --spp N [--depth D] [--grid X,Y,Z] [--arith exact|fma|fast]
                           [--check-rows R1,R2] [--save PREFIX]
Also importable: run(scene, ...) returns the result as a dict (bench.py --stress prints it as a second JSON line)."""
import argparse, os, sys, tempfile, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def run(scene="stress", res=(1920, 1080), spp=100, depth=8, grid=(22, 22, 21), arith="exact", check_rows=(), save="",
        iters_per_batch=0, debug_flags=0):
    from cosc_4397_pathtracing_raytracing_project_amd import capi, scenes
    w, h = res
    txt = {"cornell": scenes.cornell_scene_text, "sphere": scenes.sphere_scene_text}.get(scene)
    if scene.startswith("random"):  # the scene ladder's random scenes: random156 = 150 objects + the cornell box's six
        prims = int(scene[6:])
        text = scenes.random_scene_text(100 + prims, prims - 6, res=(w, h), depth=depth)
    else:
        text = txt(res=(w, h), depth=depth) if txt else scenes.stress_scene_text(tuple(grid), res=(w, h), depth=depth)
    path = scenes.write_scene(text, os.path.join(tempfile.mkdtemp(), scene + ".txt"))
    sc = capi.Scene(path, res=(w, h))
    kw = dict(iters_per_batch=iters_per_batch, time_kernels=True, debug_flags=debug_flags, arith=arith)
    r = capi.Renderer(sc, **kw)
    r.render(1, max(2, spp // 10)); r.clear()  # warm-up, then the accumulation restarts on the same buffers
    t0 = time.perf_counter(); r.render(1, spp); img = r.readback(); dt = time.perf_counter() - t0
    st = r.stats(); r.free()
    live = np.array(st.live_rays[:depth], float)
    out = {"metric": f"Msamples/s, {scene} {w}x{h} depth {depth}", "value": round(w * h * spp / dt / 1e6, 2), "unit": "Msamples/s",
           "config": {"workload": f"{scene}: {sc.desc.num_geoms} primitives, {len(sc.bvh())} BVH nodes, {w}x{h}, {spp} spp, depth {depth}",
                      "arith": arith, "traversal": f"uniform grid, {int(st.grid_cells)} cells" if st.grid_cells else "bvh", "iters_per_batch": int(st.iters_per_batch), "device_mem_mb": round(st.device_bytes / 2 ** 20)},
           "seconds": round(dt, 4), "live_rays_per_sample": round(float(live.sum() / st.samples), 4),
           "alive_by_depth": np.round(live / st.samples, 4).tolist(),
           "dominant_kernel_us": round(st.intersect_ms / max(1, st.intersect_launches) * 1e3, 1), "dominant_kernel_launches": int(st.intersect_launches),
           "mean_rgb": (img / np.float32(spp)).mean(axis=0, dtype=np.float64).tolist(), "finite": bool(np.isfinite(img).all())}
    if check_rows:
        from oracle import binding as ob
        exact = arith == "exact"
        ob.set_math_mode(ob.PORTABLE if exact else ob.LIBM); ob.load_scene(path, res=(w, h))
        rows = {}
        for row in check_rows:
            ref = ob.render(1, spp, depth=depth, variant=ob.RETIRE if exact else ob.LITERAL, nthreads=min(32, os.cpu_count()), pix_begin=row * w, pix_count=w)
            got = img[row * w:(row + 1) * w]
            if exact:
                rows[str(row)] = {"bit_exact_vs_oracle": bool(np.array_equal(got.view(np.uint32), ref.view(np.uint32)))}
            else:
                a, b = got / np.float32(spp), ref / np.float32(spp)
                mse = float(np.mean((a.astype(np.float64) - b) ** 2))
                rows[str(row)] = {"psnr_vs_reference_semantics_db": round(10 * np.log10(1 / max(mse, 1e-30)), 2),
                                  "pixels_off_by_1e-5": float((np.abs(a - b).max(axis=1) > 1e-5).mean())}
        out["rows"] = rows
    if save:
        capi.save_png(save + ".png", img, w, h, float(spp))
    return out


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("scene")
    ap.add_argument("--res", default="1920x1080")
    ap.add_argument("--spp", type=int, default=100)
    ap.add_argument("--depth", type=int, default=8)
    ap.add_argument("--grid", default="22,22,21")
    ap.add_argument("--arith", default="exact", choices=["exact", "fma", "fast"])
    ap.add_argument("--check-rows", default="")
    ap.add_argument("--save", default="")
    ap.add_argument("--iters-per-batch", type=int, default=0)
    ap.add_argument("--debug-flags", type=int, default=0)
    a = ap.parse_args()
    import json
    print(json.dumps(run(a.scene, tuple(map(int, a.res.split("x"))), a.spp, a.depth, tuple(map(int, a.grid.split(","))), a.arith,
                         [int(x) for x in a.check_rows.split(",") if x], a.save, a.iters_per_batch, a.debug_flags)))
