"""Reader for the binary goldens of oracle/ref_hot_harness.cpp / ref_rng_harness.cpp (format: oracle/ref_gold_io.h).

Data only: the fixtures hold inputs and the outputs the reference's own code (compiled in place by `make -C oracle
goldens`, in the build container) produced for them; nothing here needs /root/reference at test time."""
import gzip
import os
import struct

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name: str) -> dict:
    """{section name: uint32 array [rows, cols]} of tests/golden/<name> (.bin or .bin.gz)."""
    path = os.path.join(GOLDEN, name)
    raw = gzip.open(path, "rb").read() if path.endswith(".gz") else open(path, "rb").read()
    assert raw[:8] == b"PTGOLD01", path
    n, = struct.unpack_from("<I", raw, 8)
    pos, table = 12, []
    for _ in range(n):
        nm, rows, cols = struct.unpack_from("<24sII", raw, pos)
        table.append((nm.rstrip(b"\0").decode(), rows, cols))
        pos += 32
    out = {}
    for nm, rows, cols in table:
        out[nm] = np.frombuffer(raw, "<u4", rows * cols, pos).reshape(rows, cols).copy()
        pos += 4 * rows * cols
    assert pos == len(raw), path
    return out


def f32(a: np.ndarray) -> np.ndarray:
    return np.ascontiguousarray(a, np.uint32).view(np.float32)


def same_bits_or_both_nan(a: np.ndarray, b: np.ndarray) -> np.ndarray:
    """Elementwise: equal bit patterns, or both NaN (a NaN's sign / payload is not part of the contract)."""
    a = np.ascontiguousarray(a, np.uint32)
    b = np.ascontiguousarray(b, np.uint32)
    fa, fb = a.view(np.float32), b.view(np.float32)
    return (a == b) | (np.isnan(fa) & np.isnan(fb))


def leaf_boxes_in_visit_order(bvh):
    """[(geomIndex, bmin[3], bmax[3])] of the BVH's leaves in the order computeIntersections reaches them
    (pathtrace.cu:305-324: push left, push right, so the right child is popped first)."""
    out, stack = [], [0]
    while stack:
        nd = bvh[stack.pop()]
        if nd.geomIndex >= 0:
            out.append((int(nd.geomIndex), np.array(nd.bmin[:], np.float32), np.array(nd.bmax[:], np.float32)))
        else:
            stack.append(nd.left)
            stack.append(nd.right)
    return out


def passes_aabb(o, d, bmin, bmax):
    """intersectAABB (pathtrace.cu:113-128) in numpy float32, one box against rays o, d [n, 3] (vectorised form for
    expected_closest_hits; checked against the reference's own function on the golden rays x every node box:
    tests/test_ref_pt_goldens.py::test_intersect_aabb_matches_reference)."""
    n = len(o)
    tmin = np.zeros(n, np.float32)
    tmax = np.full(n, np.finfo(np.float32).max, np.float32)
    ok = np.ones(n, bool)
    with np.errstate(all="ignore"):
        for i in range(3):
            inv = np.float32(1.0) / d[:, i]
            t0 = (bmin[i] - o[:, i]) * inv
            t1 = (bmax[i] - o[:, i]) * inv
            swap = inv < 0
            t0, t1 = np.where(swap, t1, t0), np.where(swap, t0, t1)
            tmin = np.fmax(tmin, t0)   # fmaxf / fminf ignore a NaN operand, like np.fmax / np.fmin
            tmax = np.fmin(tmax, t1)
            ok &= ~(tmax <= tmin)
    return ok


def expected_closest_hits(gold: dict, s: int, bvh):
    """What computeIntersections returns for the rays of set s, derived from the reference-compiled per-primitive
    results: among the leaves whose box the ray passes, the smallest t > 0, ties to the leaf visited first
    (`t > 0 && t < t_min`, pathtrace.cu:314).  Returns (t bits [n], hit row [n, 8], geom [n]); geom -1 = miss."""
    ng, nr = (int(x) for x in gold["sets"][s])
    rays = f32(gold[f"rays_{s}"])
    hits = gold[f"hits_{s}"].reshape(nr, ng, 8)
    o, d = rays[:, 0:3], rays[:, 3:6]
    best_t = np.full(nr, np.inf, np.float32)
    best_g = np.full(nr, -1, np.int64)
    for g, bmin, bmax in leaf_boxes_in_visit_order(bvh):
        t = f32(hits[:, g, 0])
        take = passes_aabb(o, d, bmin, bmax) & (t > 0) & (t < best_t)
        best_t = np.where(take, t, best_t)
        best_g = np.where(take, g, best_g)
    rows = hits[np.arange(nr), np.maximum(best_g, 0)]
    return rows, best_g
