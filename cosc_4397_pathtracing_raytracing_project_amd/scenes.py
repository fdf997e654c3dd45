"""Scene-file synthesiser for the reference's text scene format.

The renderer reads the reference's ``scenes/*.txt`` unchanged (format parsed by
``src/scene.cpp:7-188``).  The GPU box has no copy of the reference, so the
benchmark / test inputs are *generated* here from a compact description rather
than shipped as copies of the reference's files.  The emitted text parses to the
same materials, objects and camera as the reference's ``scenes/cornell.txt``
(values: that file, lines 2-116) and ``scenes/sphere.txt`` (lines 2-28); config
C5 is the cornell box with the sphere replaced by a 3-D grid of
alternating cubes/spheres.
"""
from __future__ import annotations

import os
from typing import Iterable, List, Sequence, Tuple


def _material(idx: int, rgb, specex=0, specrgb=(0, 0, 0), refl=0, refr=0, refrior=0, emittance=0) -> str:
    f = lambda v: " ".join(_num(x) for x in v)
    return (
        f"MATERIAL {idx}\n"
        f"RGB         {f(rgb)}\n"
        f"SPECEX      {_num(specex)}\n"
        f"SPECRGB     {f(specrgb)}\n"
        f"REFL        {_num(refl)}\n"
        f"REFR        {_num(refr)}\n"
        f"REFRIOR     {_num(refrior)}\n"
        f"EMITTANCE   {_num(emittance)}\n\n"
    )


def _num(x) -> str:
    if isinstance(x, str):
        return x
    if float(x) == int(x):
        return str(int(x))
    return repr(float(x))


def _camera(res, fovy, iterations, depth, name, eye, lookat, up) -> str:
    f = lambda v: " ".join(_num(x) for x in v)
    return (
        "CAMERA\n"
        f"RES         {res[0]} {res[1]}\n"
        f"FOVY        {_num(fovy)}\n"
        f"ITERATIONS  {iterations}\n"
        f"DEPTH       {depth}\n"
        f"FILE        {name}\n"
        f"EYE         {f(eye)}\n"
        f"LOOKAT      {f(lookat)}\n"
        f"UP          {f(up)}\n\n"
    )


def _object(idx: int, kind: str, material: int, trans, rotat, scale) -> str:
    f = lambda v: " ".join(_num(x) for x in v)
    return (
        f"OBJECT {idx}\n"
        f"{kind}\n"
        f"material {material}\n"
        f"TRANS       {f(trans)}\n"
        f"ROTAT       {f(rotat)}\n"
        f"SCALE       {f(scale)}\n\n"
    )


_CORNELL_MATERIALS = [
    dict(rgb=(1, 1, 1), emittance=1.5),                                   # 0 light
    dict(rgb=(".98", ".98", ".98")),                                      # 1 white
    dict(rgb=(".85", ".35", ".35")),                                      # 2 red
    dict(rgb=(".35", ".85", ".35")),                                      # 3 green
    dict(rgb=(".98", ".98", ".98"), specrgb=(".98", ".98", ".98"), refl=1),  # 4 "specular white"
]

# (kind, material, TRANS, ROTAT, SCALE) — cornell.txt objects 0..6
_CORNELL_OBJECTS = [
    ("cube", 0, (0, 10, 0), (0, 0, 0), (3, ".3", 3)),        # ceiling light
    ("cube", 1, (0, 0, 0), (0, 0, 0), (10, ".01", 10)),      # floor
    ("cube", 1, (0, 10, 0), (0, 0, 90), (".01", 10, 10)),    # ceiling
    ("cube", 1, (0, 5, -5), (0, 90, 0), (".01", 10, 10)),    # back wall
    ("cube", 2, (-5, 5, 0), (0, 0, 0), (".01", 10, 10)),     # left wall
    ("cube", 3, (5, 5, 0), (0, 0, 0), (".01", 10, 10)),      # right wall
    ("sphere", 4, (-1, 4, -1), (0, 0, 0), (3, 3, 3)),        # sphere
]


def cornell_scene_text(res: Tuple[int, int] = (800, 800), iterations: int = 1000, depth: int = 8,
                       name: str = "cornell") -> str:
    """Text equivalent to the reference's scenes/cornell.txt (RES/ITERATIONS/DEPTH adjustable)."""
    out: List[str] = []
    for i, m in enumerate(_CORNELL_MATERIALS):
        out.append(_material(i, **m))
    out.append(_camera(res, 45, iterations, depth, name, ("0.0", 5, "10.5"), (0, 5, 0), (0, 1, 0)))
    for i, (kind, mat, t, r, s) in enumerate(_CORNELL_OBJECTS):
        out.append(_object(i, kind, mat, t, r, s))
    return "".join(out)


def sphere_scene_text(res: Tuple[int, int] = (800, 800), iterations: int = 5000, depth: int = 8,
                      name: str = "sphere") -> str:
    """Text equivalent to the reference's scenes/sphere.txt."""
    return (
        _material(0, rgb=(1, 1, 1), emittance=5)
        + _camera(res, 45, iterations, depth, name, ("0.0", 5, "10.5"), (0, 5, 0), (0, 1, 0))
        + _object(0, "sphere", 0, (0, 0, 0), (0, 0, 0), (3, 3, 3))
    )


def stress_scene_text(grid: Tuple[int, int, int] = (22, 22, 21), res: Tuple[int, int] = (1920, 1080),
                      iterations: int = 2000, depth: int = 8, name: str = "stress") -> str:
    """Config C5: cornell walls + light, sphere replaced by a grid of small
    alternating cubes/spheres (scale .25, materials cycling 1..4, rotated about y)."""
    out: List[str] = []
    for i, m in enumerate(_CORNELL_MATERIALS):
        out.append(_material(i, **m))
    out.append(_camera(res, 45, iterations, depth, name, ("0.0", 5, "10.5"), (0, 5, 0), (0, 1, 0)))
    idx = 0
    for kind, mat, t, r, s in _CORNELL_OBJECTS[:6]:
        out.append(_object(idx, kind, mat, t, r, s))
        idx += 1
    nx, ny, nz = grid
    pitch = 0.4
    for iz in range(nz):
        for iy in range(ny):
            for ix in range(nx):
                k = ix + nx * (iy + ny * iz)
                x = (ix - (nx - 1) / 2.0) * pitch
                y = 0.6 + iy * pitch
                z = (iz - (nz - 1) / 2.0) * pitch
                kind = "cube" if (k & 1) == 0 else "sphere"
                mat = 1 + (k % 4)
                out.append(_object(idx, kind, mat, (round(x, 4), round(y, 4), round(z, 4)),
                                   (0, (k * 7) % 90, 0), (".25", ".25", ".25")))
                idx += 1
    return "".join(out)


def random_scene_text(seed: int, n_objects: int, res: Tuple[int, int] = (96, 64), iterations: int = 16, depth: int = 8,
                      clustered: bool = False, name: str = "random") -> str:
    """Fuzz input for the parity tests: `n_objects` cubes / spheres with random translation, rotation about all three
    axes and NON-uniform scale inside (and poking through) the cornell box, random materials out of six (diffuse,
    mirror, partly reflective with / without the refractive flag, two emitters).  `clustered` puts 80 % of the
    objects into one corner so that the median-split BVH and its flattened top become unbalanced in extent.
    Plain python `random` (Mersenne twister) so that the text is identical on every machine."""
    import random
    rnd = random.Random(seed)
    mats = [
        dict(rgb=(1, 1, 1), emittance=2),
        dict(rgb=(".9", ".8", ".7")),
        dict(rgb=(".2", ".7", ".9")),
        dict(rgb=(".95", ".95", ".95"), specrgb=(".9", ".9", ".9"), refl=1),
        dict(rgb=(".8", ".3", ".3"), specrgb=(".7", ".7", ".2"), refl=".4"),
        dict(rgb=(".3", ".8", ".3"), specrgb=(".9", ".9", ".9"), refl=".6", refr=".5", refrior="1.5"),
        dict(rgb=(1, ".6", ".2"), emittance=".7"),
    ]
    out: List[str] = [_material(i, **m) for i, m in enumerate(mats)]
    out.append(_camera(res, 45, iterations, depth, name, ("0.0", 5, "10.5"), (0, 5, 0), (0, 1, 0)))
    idx = 0
    for kind, mat, t, r, sc in _CORNELL_OBJECTS[:6]:
        out.append(_object(idx, kind, min(mat, 2), t, r, sc))
        idx += 1
    for k in range(n_objects):
        if clustered and rnd.random() < 0.8:
            t = (rnd.uniform(2.5, 4.8), rnd.uniform(0.2, 2.5), rnd.uniform(-4.8, -2.5))
            smax = 0.5
        else:
            t = (rnd.uniform(-5.5, 5.5), rnd.uniform(-0.5, 10.5), rnd.uniform(-5.5, 4.0))
            smax = 2.5 if n_objects < 100 else 0.8
        r = tuple(round(rnd.uniform(0, 360), 2) for _ in range(3))
        sc = tuple(round(rnd.uniform(0.05, smax), 3) for _ in range(3))
        kind = "cube" if rnd.random() < 0.5 else "sphere"
        out.append(_object(idx, kind, rnd.randrange(len(mats)), tuple(round(v, 3) for v in t), r, sc))
        idx += 1
    return "".join(out)


def write_scene(text: str, path: str) -> str:
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    with open(path, "w") as f:
        f.write(text)
    return path


def _octahedron_tris(a: float = 0.5) -> List[Tuple[float, ...]]:
    """Eight outward-facing (counter-clockwise seen from outside) triangles of the octahedron with vertices at +-a."""
    tris = []
    for sx in (1, -1):
        for sy in (1, -1):
            for sz in (1, -1):
                v0, v1, v2 = (sx * a, 0.0, 0.0), (0.0, sy * a, 0.0), (0.0, 0.0, sz * a)
                if sx * sy * sz < 0:
                    v1, v2 = v2, v1
                tris.append(v0 + v1 + v2)
    return tris


def _mesh_object(idx: int, material: int, trans, rotat, scale, tris) -> str:
    f = lambda v: " ".join(_num(x) for x in v)
    body = "".join("TRI " + " ".join(repr(float(x)) for x in t) + "\n" for t in tris)
    return (f"OBJECT {idx}\nmesh\nmaterial {material}\nTRANS       {f(trans)}\nROTAT       {f(rotat)}\n"
            f"SCALE       {f(scale)}\n{body}\n")


def mesh_scene_text(res: Tuple[int, int] = (200, 120), iterations: int = 16, depth: int = 8, name: str = "mesh",
                    grid: int = 0) -> str:
    """EXTENSION test input (not a reference scene): cornell.txt's box and sphere plus `mesh` objects — a red
    octahedron, a mirror-material octahedron, an axis-aligned two-triangle quad facing the camera — and a cube AFTER the
    meshes, whose OBJECT id only matches if mesh objects count as one object each.  grid > 0 adds grid^3 small
    octahedra (8 triangles each), enough leaves for the global-memory / subtree path."""
    out: List[str] = []
    for i, m in enumerate(_CORNELL_MATERIALS):
        out.append(_material(i, **m))
    out.append(_camera(res, 45, iterations, depth, name, ("0.0", 5, "10.5"), (0, 5, 0), (0, 1, 0)))
    idx = 0
    for kind, mat, t, r, s in _CORNELL_OBJECTS:
        out.append(_object(idx, kind, mat, t, r, s))
        idx += 1
    octa = _octahedron_tris()
    out.append(_mesh_object(idx, 2, (2, 3, 1), (20, 30, 10), (3, 4, 3), octa)); idx += 1
    out.append(_mesh_object(idx, 4, ("-2.5", "1.5", 2), (0, 45, 0), (2, 3, 2), octa)); idx += 1
    quad = [(-0.5, -0.5, 0.0, 0.5, -0.5, 0.0, 0.5, 0.5, 0.0), (-0.5, -0.5, 0.0, 0.5, 0.5, 0.0, -0.5, 0.5, 0.0)]
    out.append(_mesh_object(idx, 3, (0, 8, -3), (0, 0, 0), (4, 2, 1), quad)); idx += 1
    out.append(_object(idx, "cube", 1, (3, 1, -2), (0, 25, 0), (2, 2, 2))); idx += 1
    for k in range(grid ** 3):
        ix, iy, iz = k % grid, (k // grid) % grid, k // (grid * grid)
        t = (round((ix - (grid - 1) / 2) * 0.9, 4), round(5.5 + iy * 0.8, 4), round(-3.5 + iz * 0.9, 4))
        out.append(_mesh_object(idx, 1 + k % 4, t, ((k * 13) % 90, (k * 7) % 90, 0), (".6", ".6", ".6"), octa)); idx += 1
    return "".join(out)
