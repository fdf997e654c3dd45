"""The tightened sphere leaf boxes of large scenes (pt_traversal_boxes, csrc/pt_api.cpp sphere_tight_box) must never
reject a ray the reference's sphere test (intersections.h:102-144, restated by the oracle and pinned by the reference-compiled
goldens) reports as a hit: the traversal may skip a leaf only when the primitive test could not hit anyway.  CPU only: the
oracle says which rays hit which sphere, the box test is intersectAABB restated in numpy (tests/golden_io.py).  The GPU side
of the same claim is bit-exact rendering of such scenes (tests/test_gpu_grid.py, test_gpu_render.py, tools/fuzz_*)."""
import ctypes as C

import numpy as np
import pytest

from cosc_4397_pathtracing_raytracing_project_amd import capi, scenes
from golden_io import passes_aabb


def _scene(tmp_path, name):
    if name == "lattice":
        text = scenes.stress_scene_text((12, 12, 10), res=(96, 64))
    elif name == "random":
        text = scenes.random_scene_text(21, 1500, res=(96, 64))
    else:
        text = scenes.random_scene_text(22, 1200, res=(96, 64), clustered=True)
    return scenes.write_scene(text, str(tmp_path / f"{name}.txt"))


def _unit(v):
    return (v / np.linalg.norm(v, axis=0, keepdims=True)).astype(np.float32)


@pytest.mark.parametrize("name", ["lattice", "random", "clustered"])
def test_tight_boxes_keep_every_sphere_hit(tmp_path, oracle, name):
    path = _scene(tmp_path, name)
    sc = capi.Scene(path)
    oracle.load_scene(path)
    boxes, tightened = sc.traversal_boxes()
    geoms = sc.geoms()
    types = np.array([g.type for g in geoms])
    spheres = np.flatnonzero(types == 0)
    assert tightened > 0.5 * len(spheres)

    # reference boxes: the tightened ones lie inside them (monotone slab test => ancestors pass as well)
    ref = np.zeros_like(boxes)
    for nd in sc.bvh():
        if nd.left < 0:
            ref[nd.geomIndex, :3], ref[nd.geomIndex, 3:] = list(nd.bmin), list(nd.bmax)
    assert np.all(boxes[:, :3] >= ref[:, :3]) and np.all(boxes[:, 3:] <= ref[:, 3:])
    cubes = types != 0
    assert np.array_equal(boxes[cubes], ref[cubes])

    rng = np.random.default_rng(7)
    lo, hi = ref[:, :3].min(axis=0), ref[:, 3:].max(axis=0)
    cam = np.array(list(sc.desc.camera.position), np.float32)
    w, h = oracle.resolution()
    rays = [oracle.generate(0, w * h)]
    # random origins all over the scene bounds, random directions
    n = 400000
    o = (lo[:, None] + rng.uniform(0, 1, (3, n)) * (hi - lo)[:, None]).astype(np.float32)
    rays.append((o, _unit(rng.normal(size=(3, n)))))
    total_hits = 0
    for o, d in rays:
        o, d = np.ascontiguousarray(o), np.ascontiguousarray(d)
        hit = oracle.intersect(o, d)
        idx = np.flatnonzero((hit["t"] > 0) & (types[np.maximum(hit["geom"], 0)] == 0))
        gi = hit["geom"][idx]
        passes = passes_aabb(o[:, idx].T, d[:, idx].T, boxes[gi, :3].T, boxes[gi, 3:].T)
        assert passes.all(), f"{(~passes).sum()} sphere hits fail the tightened leaf box"
        total_hits += len(idx)
    assert total_hits > 5000

    # Silhouette rays — the ones whose hit / miss decision is made by float rounding: from the camera, the corners of the
    # scene and random points, aimed so that the line passes the sphere's centre at 0.5 (1 + delta) object units.  Each is
    # tested against ITS sphere alone (orc_geom_test: sphereIntersectionTest on one (ray, geom) pair, the function the
    # reference-compiled golden pins), so that nearer primitives cannot hide a grazing hit.
    L = oracle.lib()
    fp = C.POINTER(C.c_float)
    L.orc_geom_test.restype = C.c_float
    m = 40
    pbuf, nbuf, out = np.zeros(3, np.float32), np.zeros(3, np.float32), C.c_int(0)
    hits = grey = 0
    for gi in spheres[rng.permutation(len(spheres))[:300]]:
        g = geoms[gi]
        M = np.array(list(g.transform), np.float64).reshape(4, 4).T  # stored column-major: M[r][c]
        Minv = np.array(list(g.inverseTransform), np.float64).reshape(4, 4).T
        corners = lo[:, None] + rng.integers(0, 2, (3, 9)) * (hi - lo)[:, None]
        inside = lo[:, None] + rng.uniform(0, 1, (3, m - 10)) * (hi - lo)[:, None]
        org = np.concatenate([cam[:, None].astype(np.float64), corners, inside], axis=1).astype(np.float32).astype(np.float64)
        ro = Minv[:3, :3] @ org + Minv[:3, 3:4]
        y = np.cross(ro.T, rng.normal(size=(m, 3))).T
        y /= np.linalg.norm(y, axis=0, keepdims=True)
        delta = rng.choice([0.0, 1e-7, -1e-7, 3e-7, 1e-6, -1e-6, 3e-6, 1e-5, 1e-4, -1e-4, 1e-3], m)
        want = 0.5 * (1 + delta)
        nr = np.linalg.norm(ro, axis=0)
        ok = nr > 1.001 * want
        q = want * nr / np.sqrt(np.maximum(nr * nr - want * want, 1e-30))  # |q|: the line ro -> q passes the centre at `want`
        target = M[:3, :3] @ (y * q) + M[:3, 3:4]
        o32 = np.ascontiguousarray(org.T.astype(np.float32))
        d32 = np.ascontiguousarray(_unit(target - org).T)
        box_ok = passes_aabb(o32, d32, boxes[gi, :3], boxes[gi, 3:])
        for r in np.flatnonzero(ok):
            t = L.orc_geom_test(int(gi), o32[r].ctypes.data_as(fp), d32[r].ctypes.data_as(fp), pbuf.ctypes.data_as(fp),
                                nbuf.ctypes.data_as(fp), C.byref(out))
            if t > 0:
                hits += 1
                grey += delta[r] > 0  # a hit although the exact line passes outside the sphere: float rounding at work
                assert box_ok[r], (int(gi), r, float(delta[r]))
    assert hits > 2000 and grey > 50, (hits, grey)
