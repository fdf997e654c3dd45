"""ctypes binding of oracle/liboracle.so — TEST INFRASTRUCTURE ONLY.

May be imported only by tests/, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of bench.py (never by the product package).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle.so")

LIBM, PORTABLE = 0, 1
LITERAL, RETIRE = 0, 1


class OrcGeom(C.Structure):
    _fields_ = [("type", C.c_int), ("materialid", C.c_int), ("transform", C.c_float * 16),
                ("inverseTransform", C.c_float * 16), ("invTranspose", C.c_float * 16)]


class OrcMaterial(C.Structure):
    _fields_ = [("color", C.c_float * 3), ("specular_exponent", C.c_float), ("specular_color", C.c_float * 3),
                ("hasReflective", C.c_float), ("hasRefractive", C.c_float), ("indexOfRefraction", C.c_float),
                ("emittance", C.c_float)]


class OrcCamera(C.Structure):
    _fields_ = [("res", C.c_int * 2), ("position", C.c_float * 3), ("lookAt", C.c_float * 3),
                ("view", C.c_float * 3), ("up", C.c_float * 3), ("right", C.c_float * 3),
                ("fov", C.c_float * 2), ("pixelLength", C.c_float * 2)]


class OrcBVHNode(C.Structure):
    _fields_ = [("bmin", C.c_float * 3), ("bmax", C.c_float * 3), ("left", C.c_int), ("right", C.c_int),
                ("geomIndex", C.c_int)]


def build(force: bool = False) -> str:
    """Compile liboracle.so with the Makefile next to this file (g++, seconds)."""
    src = os.path.join(_HERE, "pt_oracle.cpp")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "liboracle.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        fp = C.POINTER(C.c_float)
        ip = C.POINTER(C.c_int)
        L.orc_scene_load.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int]
        L.orc_scene_load.restype = C.c_int
        L.orc_scene_set.argtypes = [C.POINTER(OrcGeom), C.c_int, C.POINTER(OrcMaterial), C.c_int,
                                    C.POINTER(OrcCamera), C.c_int]
        L.orc_image_name.restype = C.c_char_p
        L.orc_utilhash.argtypes = [C.c_uint32]
        L.orc_utilhash.restype = C.c_uint32
        L.orc_seed.argtypes = [C.c_int, C.c_int, C.c_int]
        L.orc_seed.restype = C.c_int32
        L.orc_rng_draws.argtypes = [C.c_int32, C.c_int, C.POINTER(C.c_uint32), fp]
        L.orc_minstd_nth.argtypes = [C.c_uint32, C.c_int]
        L.orc_minstd_nth.restype = C.c_uint32
        L.orc_geom_test.argtypes = [C.c_int, fp, fp, fp, fp, ip]
        L.orc_geom_test.restype = C.c_float
        L.orc_generate.argtypes = [C.c_int, C.c_int, fp, fp]
        L.orc_generate_iter.argtypes = [C.c_int, C.c_int, C.c_int, fp, fp]
        L.orc_set_aa_jitter.argtypes = [C.c_int]
        L.orc_intersect.argtypes = [C.c_int, fp, fp, fp, fp, ip, fp, ip, ip, C.POINTER(C.c_long)]
        L.orc_shade.argtypes = [C.c_int, C.c_int, ip, ip, fp, fp, ip, fp, fp, fp, fp, ip]
        L.orc_render.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, fp,
                                 C.POINTER(C.c_long)]
        L.orc_math_eval_f.argtypes = [C.c_int, C.c_int, fp, fp]
        L.orc_math_eval_d.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double)]
        L.orc_aabb_all_nodes.argtypes = [C.c_int, fp, fp, C.POINTER(C.c_uint32)]
        L.orc_helpers.argtypes = [C.c_int, C.c_int, fp, fp]
        L.orc_build_xform.argtypes = [fp, fp, fp, fp]
        L.orc_vecops.argtypes = [fp, fp, fp, fp]
        _lib = L
    return _lib


def _fp(a: np.ndarray):
    assert a.dtype == np.float32 and a.flags.c_contiguous
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _ip(a: np.ndarray):
    assert a.dtype == np.int32 and a.flags.c_contiguous
    return a.ctypes.data_as(C.POINTER(C.c_int))


def set_math_mode(mode: int) -> None:
    lib().orc_set_math_mode(mode)


def load_scene(path: str, res=None, fixup: bool = True) -> None:
    w, h = res if res else (0, 0)
    rc = lib().orc_scene_load(path.encode(), int(w), int(h), 1 if fixup else 0)
    if rc != 0:
        raise FileNotFoundError(path)


def camera() -> OrcCamera:
    c = OrcCamera()
    lib().orc_get_camera(C.byref(c))
    return c


def geoms():
    n = lib().orc_num_geoms()
    arr = (OrcGeom * n)()
    lib().orc_get_geoms(arr)
    return arr


def materials():
    n = lib().orc_num_materials()
    arr = (OrcMaterial * n)()
    lib().orc_get_materials(arr)
    return arr


def bvh():
    n = lib().orc_num_bvh_nodes()
    arr = (OrcBVHNode * n)()
    lib().orc_get_bvh(arr)
    return arr


def trace_depth() -> int:
    return lib().orc_trace_depth()


def resolution():
    c = camera()
    return c.res[0], c.res[1]


def set_aa_jitter(on: bool) -> None:
    """Extension (not in the reference): stochastic anti-aliasing of the camera rays; off = reference semantics."""
    lib().orc_set_aa_jitter(1 if on else 0)


def generate(pix_begin: int, count: int, iteration: int = 1):
    o = np.zeros((3, count), np.float32)
    d = np.zeros((3, count), np.float32)
    lib().orc_generate_iter(int(iteration), pix_begin, count, _fp(o), _fp(d))
    return o, d


def intersect(o: np.ndarray, d: np.ndarray, stats: bool = False):
    n = o.shape[1]
    t = np.zeros(n, np.float32)
    nrm = np.zeros((3, n), np.float32)
    mat = np.zeros(n, np.int32)
    pt = np.zeros((3, n), np.float32)
    geom = np.zeros(n, np.int32)
    outside = np.zeros(n, np.int32)
    st = (C.c_long * 3)()
    lib().orc_intersect(n, _fp(np.ascontiguousarray(o)), _fp(np.ascontiguousarray(d)), _fp(t), _fp(nrm), _ip(mat),
                        _fp(pt), _ip(geom), _ip(outside), st if stats else None)
    out = dict(t=t, nrm=nrm, mat=mat, pt=pt, geom=geom, outside=outside)
    if stats:
        out["stats"] = dict(node_pops=st[0], prim_tests=st[1], max_stack=st[2])
    return out


def shade(depth: int, it: np.ndarray, pixel: np.ndarray, hit: dict, o, d, color, remaining):
    """One shadeAndExtendRays step; returns new (o, d, color, remaining)."""
    n = o.shape[1]
    o, d, color = (np.ascontiguousarray(a, np.float32).copy() for a in (o, d, color))
    remaining = np.ascontiguousarray(remaining, np.int32).copy()
    lib().orc_shade(n, depth, _ip(np.ascontiguousarray(it, np.int32)), _ip(np.ascontiguousarray(pixel, np.int32)),
                    _fp(hit["t"]), _fp(hit["nrm"]), _ip(hit["mat"]), _fp(hit["pt"]), _fp(o), _fp(d), _fp(color),
                    _ip(remaining))
    return o, d, color, remaining


def render(iter_first: int, iter_count: int, depth: int = 0, variant: int = LITERAL, nthreads: int = 1,
           pix_begin: int = 0, pix_count: int | None = None, accum: np.ndarray | None = None,
           want_stats: bool = False):
    """Returns the running SUM image [pix_count, 3] float32 (like scene->state.image)."""
    w, h = resolution()
    if pix_count is None:
        pix_count = w * h - pix_begin
    img = accum if accum is not None else np.zeros((pix_count, 3), np.float32)
    st = (C.c_long * 67)() if want_stats else None
    lib().orc_render(iter_first, iter_count, depth, variant, nthreads, pix_begin, pix_count, _fp(img), st)
    if want_stats:
        return img, dict(live=list(st[0:64]), node_pops=st[64], prim_tests=st[65], max_stack=st[66])
    return img


def math_eval_f(fn: int, x: np.ndarray) -> np.ndarray:
    x = np.ascontiguousarray(x, np.float32)
    out = np.empty_like(x)
    lib().orc_math_eval_f(fn, x.size, _fp(x), _fp(out))
    return out


def math_eval_d(fn: int, x: np.ndarray) -> np.ndarray:
    x = np.ascontiguousarray(x, np.float64)
    out = np.empty_like(x)
    dp = C.POINTER(C.c_double)
    lib().orc_math_eval_d(fn, x.size, x.ctypes.data_as(dp), out.ctypes.data_as(dp))
    return out


def build_xform(trs):
    t = np.ascontiguousarray(trs, np.float32)
    m, i, it = (np.zeros(16, np.float32) for _ in range(3))
    lib().orc_build_xform(_fp(t), _fp(m), _fp(i), _fp(it))
    return m, i, it


def vecops(a, b, m16):
    out = np.zeros(14, np.float32)
    lib().orc_vecops(_fp(np.ascontiguousarray(a, np.float32)), _fp(np.ascontiguousarray(b, np.float32)),
                     _fp(np.ascontiguousarray(m16, np.float32)), _fp(out))
    return out


def aabb_all_nodes(o: np.ndarray, d: np.ndarray) -> np.ndarray:
    """intersectAABB of rays o, d [n, 3] against every node box of the loaded scene: uint32 [n, ceil(nodes / 32)] bit rows."""
    o = np.ascontiguousarray(o, np.float32)
    d = np.ascontiguousarray(d, np.float32)
    words = (lib().orc_num_bvh_nodes() + 31) // 32
    out = np.zeros((len(o), words), np.uint32)
    lib().orc_aabb_all_nodes(len(o), _fp(o), _fp(d), out.ctypes.data_as(C.POINTER(C.c_uint32)))
    return out


def helpers(kind: int, inputs: np.ndarray) -> np.ndarray:
    """The sampling helpers of pathtrace.cu:216-242 in the current math mode (0 frame, 1 cosine-weighted sample, 2 reflect)."""
    x = np.ascontiguousarray(inputs, np.float32)
    out = np.zeros((len(x), 6 if kind == 0 else 3), np.float32)
    lib().orc_helpers(kind, len(x), _fp(x), _fp(out))
    return out
