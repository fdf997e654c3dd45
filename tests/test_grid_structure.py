"""The traversal structure of our own for large scenes: a uniform grid over the leaf boxes
(pt_build_grid, include/pt_amd.h).  CPU-only checks of the two facts the GPU walk's bit-exactness rests on:
  * the reference's float AABB test (pathtrace.cu:113-128) is monotone under box inclusion — a ray that passes a leaf's
    box passes the boxes of all its ancestors — so "the leaves the reference's walk tests" == "the leaves whose own box the
    ray passes", whatever structure finds them;
  * every leaf is listed in every cell its box (grown by the grid's pad) touches, records carry the leaf's exact box and
    visiting rank, and the neighbour bits the walk uses to skip repeats say where else the leaf is listed.
The GPU side (tests/test_gpu_render.py, forced-grid cases and the C5 rows) compares images bit for bit."""
import numpy as np
import pytest

from cosc_4397_pathtracing_raytracing_project_amd import capi, scenes

F = np.float32


def ref_aabb_pass(bmin, bmax, o, d):
    """intersectAABB of the reference for one ray against many boxes, float32 operation by operation (fmaxf / fminf ignore
    a NaN operand, np.fmax / np.fmin do the same); the early return per axis is equivalent to one test at the end
    because tmin only grows and tmax only shrinks."""
    tmin = np.zeros(len(bmin), F)
    tmax = np.full(len(bmin), np.finfo(F).max, F)
    dead = np.zeros(len(bmin), bool)
    with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
        for a in range(3):
            inv = F(1.0) / d[a]
            t0 = (bmin[:, a] - o[a]) * inv
            t1 = (bmax[:, a] - o[a]) * inv
            if inv < 0:
                t0, t1 = t1, t0
            tmin = np.fmax(tmin, t0)
            tmax = np.fmin(tmax, t1)
            dead |= tmax <= tmin
    return ~dead


def scene_nodes(path, res=(96, 64)):
    sc = capi.Scene(path, res=res)
    B = sc.bvh()
    bmin = np.array([[b.bmin[0], b.bmin[1], b.bmin[2]] for b in B], F)
    bmax = np.array([[b.bmax[0], b.bmax[1], b.bmax[2]] for b in B], F)
    left = np.array([b.left for b in B])
    right = np.array([b.right for b in B])
    geom = np.array([b.geomIndex for b in B])
    parent = np.full(len(B), -1)
    for i in range(len(B)):
        if left[i] >= 0:
            parent[left[i]] = parent[right[i]] = i
    return sc, bmin, bmax, parent, geom


@pytest.fixture(scope="module")
def test_scenes(tmp_path_factory):
    d = tmp_path_factory.mktemp("grid")
    return {
        "lattice": scenes.write_scene(scenes.stress_scene_text((10, 10, 8), res=(96, 64)), str(d / "lattice.txt")),
        "random": scenes.write_scene(scenes.random_scene_text(4, 300, res=(96, 64)), str(d / "random.txt")),
        "clustered": scenes.write_scene(scenes.random_scene_text(6, 600, res=(96, 64), clustered=True), str(d / "clustered.txt")),
        "mesh": scenes.write_scene(scenes.mesh_scene_text(res=(96, 64), grid=4), str(d / "mesh.txt")),
    }


@pytest.mark.parametrize("name", ["lattice", "random", "clustered", "mesh"])
def test_reference_box_test_is_monotone_under_inclusion(test_scenes, name):
    sc, bmin, bmax, parent, geom = scene_nodes(test_scenes[name])
    # parents are exact unions of their children (min / max of floats): the premise of the argument
    for i in np.flatnonzero(parent >= 0):
        assert (bmin[parent[i]] <= bmin[i]).all() and (bmax[parent[i]] >= bmax[i]).all()
    rng = np.random.default_rng(7)
    lo, hi = bmin[0], bmax[0]
    leaves = np.flatnonzero(geom >= 0)
    rays = []
    for k in range(400):
        o = (lo + (hi - lo) * rng.random(3) * 1.2 - 0.1 * (hi - lo)).astype(F)
        d = rng.normal(size=3)
        d = (d / np.linalg.norm(d)).astype(F)
        kind = k % 8
        if kind == 1:    # axis-parallel: a zero component (1/0 = inf, and 0 * inf = NaN on a box plane)
            d[rng.integers(3)] = F(0.0)
        elif kind == 2:  # two zero components
            z = rng.permutation(3)[:2]
            d[z] = F(0.0)
        elif kind == 3:  # negative zero
            d[rng.integers(3)] = F(-0.0)
        if kind in (4, 5) or kind in (1, 2, 3) and k % 16 < 8:
            # origin exactly on planes of a leaf box (where the bounce rays of axis-aligned surfaces start)
            leaf = leaves[rng.integers(len(leaves))]
            ax = rng.integers(3)
            o[ax] = (bmin if rng.random() < 0.5 else bmax)[leaf, ax]
        if not np.any(d != 0):
            d[0] = F(1.0)
        rays.append((o, d))
    checked = 0
    for o, d in rays:
        ok = ref_aabb_pass(bmin, bmax, o, d)
        for leaf in leaves[ok[leaves]]:
            i = parent[leaf]
            while i >= 0:
                assert ok[i], f"ray {o} {d}: leaf {leaf} passes, ancestor {i} does not"
                i = parent[i]
            checked += 1
    assert checked > 100


@pytest.mark.parametrize("name", ["lattice", "random", "clustered", "mesh"])
def test_grid_lists_every_leaf_in_every_cell_its_box_touches(test_scenes, name):
    sc, bmin, bmax, parent, geom = scene_nodes(test_scenes[name])
    if len(geom) < 600:  # pt_api.cpp kGridNodes
        assert sc.grid() is None, "small scenes keep the BVH scan unless forced"
    info, start, recs = sc.grid(forced=True)
    res = np.array(info.res)
    origin = np.array(info.origin, np.float64)
    cs = np.array(info.cell_size, np.float64)
    n_leaves = int((geom >= 0).sum())
    assert info.num_leaves == n_leaves and info.num_cells == res.prod() and len(start) == info.num_cells + 1
    assert start[0] == 0 and start[-1] == info.num_records and (np.diff(start.astype(np.int64)) >= 0).all()
    # the grid covers the tree's bounds with room to spare
    assert (origin < bmin[0] - info.pad).all() and (origin + cs * res > bmax[0] + info.pad).all()
    rb = np.array([[r.bmin[0], r.bmin[1], r.bmin[2], r.bmax[0], r.bmax[1], r.bmax[2]] for r in recs], F)
    gtype = np.array([g.type for g in sc.geoms()])
    gbox = np.zeros((len(gtype), 6), F)  # leaf box of every primitive, from the reference tree
    for i in np.flatnonzero(geom >= 0):
        gbox[geom[i]] = np.concatenate([bmin[i], bmax[i]])
    rleaf = np.array([r.leaf for r in recs])
    rbits = np.array([r.neighbours for r in recs])
    cell_of = np.repeat(np.arange(info.num_cells), np.diff(start.astype(np.int64)))
    # records carry exact leaf boxes: as a multiset they are the reference tree's leaf boxes
    leaf_boxes = np.concatenate([bmin[geom >= 0], bmax[geom >= 0]], axis=1)
    first = {}
    for k in range(len(recs)):
        first.setdefault(int(rleaf[k]), k)
    assert len(first) == n_leaves
    got = np.array(sorted(map(tuple, rb[list(first.values())])))
    want = np.array(sorted(map(tuple, leaf_boxes)))
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    members = {}
    for k in range(len(recs)):
        members.setdefault(int(rleaf[k]), set()).add(int(cell_of[k]))
        assert np.array_equal(rb[k].view(np.uint32), rb[first[int(rleaf[k])]].view(np.uint32))
    for leaf, cells in members.items():
        b = rb[first[leaf]].astype(np.float64)
        # every cell the box grown by HALF the pad touches must be listed (the builder uses the whole pad; half leaves
        # room for the float cell arithmetic of this check), and nothing beyond the box grown by TWICE the pad
        c0 = np.clip(np.floor((b[:3] - 0.5 * info.pad - origin) / cs).astype(int), 0, res - 1)
        c1 = np.clip(np.floor((b[3:] + 0.5 * info.pad - origin) / cs).astype(int), 0, res - 1)
        need = {x + res[0] * (y + res[1] * z) for z in range(c0[2], c1[2] + 1) for y in range(c0[1], c1[1] + 1)
                for x in range(c0[0], c1[0] + 1)}
        assert need <= cells
        d0 = np.clip(np.floor((b[:3] - 2 * info.pad - origin) / cs).astype(int), 0, res - 1)
        d1 = np.clip(np.floor((b[3:] + 2 * info.pad - origin) / cs).astype(int), 0, res - 1)
        assert len(cells) <= np.prod(d1 - d0 + 1)
    # neighbour bits: bit a set <=> the leaf is also listed in the neighbour cell (-x, +x, -y, +y, -z, +z)
    step = [(-1, 0, 0), (1, 0, 0), (0, -1, 0), (0, 1, 0), (0, 0, -1), (0, 0, 1)]
    for k in range(0, len(recs), max(1, len(recs) // 4000)):
        c = int(cell_of[k])
        x, y, z = c % res[0], (c // res[0]) % res[1], c // (res[0] * res[1])
        for a, (dx, dy, dz) in enumerate(step):
            nx, ny, nz = x + dx, y + dy, z + dz
            inside = 0 <= nx < res[0] and 0 <= ny < res[1] and 0 <= nz < res[2]
            listed = inside and (nx + res[0] * (ny + res[1] * nz)) in members[int(rleaf[k])]
            assert bool((rbits[k] >> a) & 1) == listed
        gi = rbits[k] >> 8  # the primitive behind the leaf and its type ride along (one fetch less per candidate)
        assert (rbits[k] >> 6) & 3 == gtype[gi]
        assert np.array_equal(rb[k].view(np.uint32), gbox[gi].view(np.uint32))
    # records of a cell are in the reference's visiting order
    for c in range(0, info.num_cells, max(1, info.num_cells // 500)):
        seg = rleaf[start[c]:start[c + 1]]
        assert (np.diff(seg) > 0).all()


def test_grid_candidates(tmp_path):
    even = capi.Scene(scenes.write_scene(scenes.stress_scene_text((12, 12, 10), res=(96, 64)), str(tmp_path / "e.txt")))
    assert len(even.bvh()) >= 2048
    info, start, recs = even.grid()
    # a lattice of primitives: the resolution search lands on the lattice pitch (few references per primitive)
    assert info.num_records < 5 * info.num_leaves
    clustered = capi.Scene(scenes.write_scene(scenes.random_scene_text(6, 1500, res=(96, 64), clustered=True), str(tmp_path / "c.txt")))
    assert len(clustered.bvh()) >= 2048
    g = clustered.grid()
    if g is not None:  # a candidate only within the bound the builder promises (the renderer then measures both)
        assert g[0].num_records <= 64 * g[0].num_leaves
