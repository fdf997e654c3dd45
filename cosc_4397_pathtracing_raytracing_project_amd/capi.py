"""ctypes binding of libpt_amd.so (the C ABI declared in include/pt_amd.h).

This is plumbing for the Python-side drivers (bench.py, tests, multi-GPU launcher);
all rendering happens in the HIP library.  There is NO CPU fallback: if the shared
library is missing, or no HIP device is present when a render entry point is
called, an exception is raised.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional, Sequence, Tuple

import numpy as np

_PKG = os.path.dirname(os.path.abspath(__file__))
# PT_AMD_LIB lets experiment tooling (tools/ab*.sh, tools/pmc_*.sh) load an A/B build from elsewhere instead of
# overwriting the in-tree product library.
LIB_PATH = os.environ.get("PT_AMD_LIB") or os.path.join(_PKG, "libpt_amd.so")
PT_MAX_DEPTH = 64
ARITH = {"exact": 0, "fma": 1, "fast": 2}  # PT_ARITH_* (include/pt_amd.h)
ARITH_NAMES = {v: k for k, v in ARITH.items()}


class PtGeom(C.Structure):
    _fields_ = [("type", C.c_int32), ("materialid", C.c_int32), ("transform", C.c_float * 16),
                ("inverseTransform", C.c_float * 16), ("invTranspose", C.c_float * 16)]


class PtMaterial(C.Structure):
    _fields_ = [("color", C.c_float * 3), ("specular_exponent", C.c_float), ("specular_color", C.c_float * 3),
                ("hasReflective", C.c_float), ("hasRefractive", C.c_float), ("indexOfRefraction", C.c_float),
                ("emittance", C.c_float)]


class PtCamera(C.Structure):
    _fields_ = [("resolution", C.c_int32 * 2), ("position", C.c_float * 3), ("lookAt", C.c_float * 3),
                ("view", C.c_float * 3), ("up", C.c_float * 3), ("right", C.c_float * 3), ("fov", C.c_float * 2),
                ("pixelLength", C.c_float * 2)]


class PtBVHNode(C.Structure):
    _fields_ = [("bmin", C.c_float * 3), ("bmax", C.c_float * 3), ("left", C.c_int32), ("right", C.c_int32),
                ("geomIndex", C.c_int32)]


class PtGridInfo(C.Structure):
    _fields_ = [("res", C.c_int32 * 3), ("origin", C.c_float * 3), ("cell_size", C.c_float * 3), ("pad", C.c_float),
                ("num_cells", C.c_int32), ("num_records", C.c_int32), ("num_leaves", C.c_int32)]


class PtGridRecord(C.Structure):
    _fields_ = [("bmin", C.c_float * 3), ("bmax", C.c_float * 3), ("leaf", C.c_int32), ("neighbours", C.c_int32)]


class PtSceneDesc(C.Structure):
    _fields_ = [("geoms", C.POINTER(PtGeom)), ("num_geoms", C.c_int32), ("materials", C.POINTER(PtMaterial)),
                ("num_materials", C.c_int32), ("camera", PtCamera), ("trace_depth", C.c_int32)]


class PtOptions(C.Structure):
    _fields_ = [("device", C.c_int32), ("pixel_begin", C.c_int32), ("pixel_count", C.c_int32),
                ("iters_per_batch", C.c_int32), ("num_queues", C.c_int32), ("blocks_per_cu", C.c_int32),
                ("time_kernels", C.c_int32), ("legacy_traversal", C.c_int32), ("debug_flags", C.c_int32), ("unfused_primary", C.c_int32), ("unfused_bounces", C.c_int32), ("stripe_pixels", C.c_int32), ("stripe_stride", C.c_int32),
                ("arith", C.c_int32), ("aa_jitter", C.c_int32), ("reserved", C.c_int32 * 1)]


class PtStats(C.Structure):
    _fields_ = [("samples", C.c_int64), ("live_rays", C.c_int64 * PT_MAX_DEPTH), ("intersect_launches", C.c_int64),
                ("intersect_ms", C.c_double), ("render_ms", C.c_double), ("num_cus", C.c_int32),
                ("grid_blocks", C.c_int32), ("num_queues", C.c_int32), ("iters_per_batch", C.c_int32),
                ("device_bytes", C.c_int64), ("primary_fused", C.c_int32), ("bounces_fused", C.c_int32),
                ("arith", C.c_int32), ("grid_cells", C.c_int32), ("tight_leaves", C.c_int32), ("paths_waves", C.c_int32)]


class PtError(RuntimeError):
    pass


_lib: Optional[C.CDLL] = None
_fp = C.POINTER(C.c_float)
_ip = C.POINTER(C.c_int32)


def lib() -> C.CDLL:
    """Load libpt_amd.so; fail loudly when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise PtError(f"{LIB_PATH} not found: run `python -c 'import __graft_entry__ as g; g.build()'` "
                      "(hipcc --offload-arch=gfx950).  There is no CPU fallback.")
    L = C.CDLL(LIB_PATH)
    L.pt_last_error.restype = C.c_char_p
    L.pt_scene_load.argtypes = [C.c_char_p, C.c_int, C.c_int, C.POINTER(C.c_void_p)]
    L.pt_scene_free.argtypes = [C.c_void_p]
    L.pt_scene_free.restype = None
    L.pt_scene_desc.argtypes = [C.c_void_p, C.POINTER(PtSceneDesc)]
    L.pt_scene_iterations.argtypes = [C.c_void_p]
    L.pt_scene_image_name.argtypes = [C.c_void_p]
    L.pt_scene_image_name.restype = C.c_char_p
    L.pt_build_bvh.argtypes = [C.POINTER(PtGeom), C.c_int, C.POINTER(PtBVHNode), C.c_int]
    L.pt_build_grid.argtypes = [C.POINTER(PtGeom), C.c_int, C.c_int, C.POINTER(PtGridInfo), C.POINTER(C.c_uint32), C.POINTER(PtGridRecord)]
    if hasattr(L, "pt_selfcheck_ieee"):
        L.pt_selfcheck_ieee.argtypes = [C.c_int, C.c_int, C.c_uint64, C.c_uint64, C.c_uint32, C.POINTER(C.c_uint64)]
    if hasattr(L, "pt_traversal_boxes"):  # absent from older A/B builds of the library (tools/build_rev.sh)
        L.pt_traversal_boxes.argtypes = [C.POINTER(PtGeom), C.c_int, _fp, _fp]
    if hasattr(L, "pt_center_half_box"):
        L.pt_center_half_box.argtypes = [_fp, _fp, C.c_int, _fp, _fp]
        L.pt_center_half_box.restype = None
    L.pt_build_transform.argtypes = [_fp, _fp, _fp, _fp]
    L.pt_init.argtypes = [C.POINTER(PtSceneDesc), C.POINTER(PtOptions)]
    L.pt_render.argtypes = [C.c_int, C.c_int]
    L.pt_readback.argtypes = [_fp]
    L.pt_readback_device.argtypes = [C.c_void_p]
    L.pt_preview_rgba8.argtypes = [C.c_int, C.POINTER(C.c_uint8)]
    L.pt_preview_rgba8_device.argtypes = [C.c_int, C.c_void_p]
    L.pt_get_stats.argtypes = [C.POINTER(PtStats)]
    L.pt_stage_generate.argtypes = [C.c_int, C.c_int, _fp, _fp]
    L.pt_stage_intersect.argtypes = [C.c_int, _fp, _fp, _fp, _fp, _ip, _fp]
    L.pt_stage_shade.argtypes = [C.c_int, C.c_int, _ip, _ip, _fp, _fp, _ip, _fp, _fp, _fp, _fp, _ip]
    _u8p = C.POINTER(C.c_uint8)
    L.pt_save_u8.argtypes = [C.c_float, _u8p]
    L.pt_stage_save_u8.argtypes = [C.c_int, C.c_int, C.c_float, _fp, _u8p]
    L.pt_ctx_create.argtypes = [C.POINTER(PtSceneDesc), C.POINTER(PtOptions), C.POINTER(C.c_void_p)]
    L.pt_ctx_destroy.argtypes = [C.c_void_p]
    L.pt_ctx_render.argtypes = [C.c_void_p, C.c_int, C.c_int]
    L.pt_ctx_sync.argtypes = [C.c_void_p]
    L.pt_ctx_readback.argtypes = [C.c_void_p, _fp]
    L.pt_ctx_readback_device.argtypes = [C.c_void_p, C.c_void_p]
    L.pt_ctx_save_u8.argtypes = [C.c_void_p, C.c_float, _u8p]
    L.pt_ctx_get_stats.argtypes = [C.c_void_p, C.POINTER(PtStats)]
    L.pt_ctx_reset_stats.argtypes = [C.c_void_p]
    L.pt_ctx_clear.argtypes = [C.c_void_p]
    L.pt_ctx_pixel_count.argtypes = [C.c_void_p]
    L.pt_group_create.argtypes = [C.POINTER(PtSceneDesc), C.POINTER(PtOptions), _ip, C.c_int, C.POINTER(C.c_void_p)]
    if hasattr(L, "pt_group_create_ex"):  # absent only in older A/B builds loaded through PT_AMD_LIB (tools/build_rev.sh)
        L.pt_group_create_ex.argtypes = [C.POINTER(PtSceneDesc), C.POINTER(PtOptions), _ip, C.c_int, C.c_int, C.POINTER(C.c_void_p)]
        L.pt_group_transport.argtypes = [C.c_void_p]
    L.pt_group_destroy.argtypes = [C.c_void_p]
    L.pt_group_size.argtypes = [C.c_void_p]
    L.pt_group_context.argtypes = [C.c_void_p, C.c_int]
    L.pt_group_context.restype = C.c_void_p
    L.pt_group_render.argtypes = [C.c_void_p, C.c_int, C.c_int]
    L.pt_group_sync.argtypes = [C.c_void_p]
    L.pt_group_gather.argtypes = [C.c_void_p, _fp]
    L.pt_group_gather_u8.argtypes = [C.c_void_p, C.c_float, _u8p]
    L.pt_group_preview_rgba8.argtypes = [C.c_void_p, C.c_int, _u8p]
    L.pt_write_png_rgb8.argtypes = [C.c_char_p, _u8p, C.c_int, C.c_int]
    L.pt_output_basename.argtypes = [C.c_char_p, C.c_int, C.c_char_p, C.c_int]
    L.pt_save_png.argtypes = [C.c_char_p, _fp, C.c_int, C.c_int, C.c_float]
    L.pt_save_hdr.argtypes = [C.c_char_p, _fp, C.c_int, C.c_int, C.c_float]
    L.pt_save_pfm.argtypes = [C.c_char_p, _fp, C.c_int, C.c_int, C.c_float]
    _lib = L
    return L


def _check(rc: int) -> None:
    if rc != 0:
        raise PtError(lib().pt_last_error().decode(errors="replace"))


def _f(a: np.ndarray):
    assert a.dtype == np.float32 and a.flags.c_contiguous
    return a.ctypes.data_as(_fp)


def _i(a: np.ndarray):
    assert a.dtype == np.int32 and a.flags.c_contiguous
    return a.ctypes.data_as(_ip)


class Scene:
    """Host scene: `new Scene(file)` + main.cpp's initial camera state (include/pt_amd.h: pt_scene_load)."""

    def __init__(self, path: str, res: Optional[Tuple[int, int]] = None):
        self._h = C.c_void_p()
        w, h = res if res else (0, 0)
        _check(lib().pt_scene_load(os.fsencode(path), int(w), int(h), C.byref(self._h)))
        self.desc = PtSceneDesc()
        _check(lib().pt_scene_desc(self._h, C.byref(self.desc)))

    def __del__(self):
        h = getattr(self, "_h", None)
        if h is not None and h.value and _lib is not None:
            _lib.pt_scene_free(h)
            self._h = C.c_void_p()

    @property
    def resolution(self) -> Tuple[int, int]:
        return self.desc.camera.resolution[0], self.desc.camera.resolution[1]

    @property
    def trace_depth(self) -> int:
        return self.desc.trace_depth

    @trace_depth.setter
    def trace_depth(self, d: int) -> None:
        self.desc.trace_depth = int(d)

    @property
    def iterations(self) -> int:
        return lib().pt_scene_iterations(self._h)

    @property
    def image_name(self) -> str:
        return lib().pt_scene_image_name(self._h).decode()

    def geoms(self):
        return [self.desc.geoms[i] for i in range(self.desc.num_geoms)]

    def materials(self):
        return [self.desc.materials[i] for i in range(self.desc.num_materials)]

    def grid(self, forced: bool = False):
        """The uniform grid over the leaf boxes the renderer would walk (pt_build_grid): (info, cell_start, records) or
        None when the scene keeps the BVH scan."""
        info = PtGridInfo()
        rc = lib().pt_build_grid(self.desc.geoms, self.desc.num_geoms, int(forced), C.byref(info), None, None)
        if rc < 0:
            raise PtError(lib().pt_last_error().decode(errors="replace"))
        if rc == 0:
            return None
        start = (C.c_uint32 * (info.num_cells + 1))()
        recs = (PtGridRecord * info.num_records)()
        lib().pt_build_grid(self.desc.geoms, self.desc.num_geoms, int(forced), C.byref(info), start, recs)
        return info, np.frombuffer(start, np.uint32).copy(), recs

    def traversal_boxes(self):
        """(boxes [num_geoms, 6], tightened count): the leaf boxes the traversal structures of a LARGE scene test
        (pt_traversal_boxes; sphere leaves tightened to the ellipsoid's box for ray origins in the scene or at the camera)."""
        boxes = np.zeros((self.desc.num_geoms, 6), np.float32)
        cam = np.asarray(list(self.desc.camera.position), np.float32)
        rc = lib().pt_traversal_boxes(self.desc.geoms, self.desc.num_geoms, _f(cam), _f(boxes))
        if rc < 0:
            raise PtError(lib().pt_last_error().decode(errors="replace"))
        return boxes, rc

    def bvh(self):
        n = lib().pt_build_bvh(self.desc.geoms, self.desc.num_geoms, None, 0)
        arr = (PtBVHNode * n)()
        lib().pt_build_bvh(self.desc.geoms, self.desc.num_geoms, arr, n)
        return arr


def selfcheck_ieee(kind: int, first: int, count: int, seed: int = 0, arith: str = "exact") -> int:
    """Mismatches between the guarded IEEE sqrt / reciprocal / quotient sequences and the compiler's expansions on the GPU
    (pt_selfcheck_ieee)."""
    n = C.c_uint64(0)
    _check(lib().pt_selfcheck_ieee(ARITH[arith], kind, first, count, seed, C.byref(n)))
    return int(n.value)


def build_transform(trs: Sequence[float]):
    t = np.asarray(trs, np.float32).copy()
    m, i, it = (np.zeros(16, np.float32) for _ in range(3))
    lib().pt_build_transform(_f(t), _f(m), _f(i), _f(it))
    return m, i, it


def make_options(device: int = 0, pixel_begin: int = 0, pixel_count: int = 0, iters_per_batch: int = 0,
                 num_queues: int = 0, blocks_per_cu: int = 0, time_kernels: bool = False, legacy_traversal: bool = False,
                 debug_flags: int = 0, unfused_primary: bool = False, unfused_bounces: bool = False,
                 stripe_pixels: int = 0, stripe_stride: int = 0, arith="exact", aa_jitter: bool = False) -> PtOptions:
    opt = PtOptions()
    opt.device = device
    opt.pixel_begin = pixel_begin
    opt.pixel_count = pixel_count
    opt.iters_per_batch = iters_per_batch
    opt.num_queues = num_queues
    opt.blocks_per_cu = blocks_per_cu
    opt.time_kernels = 1 if time_kernels else 0
    opt.legacy_traversal = 1 if legacy_traversal else 0
    opt.debug_flags = int(debug_flags)
    opt.unfused_primary = 1 if unfused_primary else 0
    opt.unfused_bounces = 1 if unfused_bounces else 0
    opt.stripe_pixels = int(stripe_pixels)
    opt.stripe_stride = int(stripe_stride)
    opt.arith = ARITH[arith] if isinstance(arith, str) else int(arith)
    opt.aa_jitter = 1 if aa_jitter else 0
    return opt


class Renderer:
    """pathtraceInit / pathtrace / pathtraceFree over the C ABI (the default instance, like the reference's
    file-scope renderer state).  `arith`: "exact" (bit-identical to the oracle), "fma" or "fast" (PT_ARITH_*)."""

    def __init__(self, scene: Scene, device: int = 0, pixel_begin: int = 0, pixel_count: int = 0, **kw):
        opt = make_options(device=device, pixel_begin=pixel_begin, pixel_count=pixel_count, **kw)
        self.scene = scene
        w, h = scene.resolution
        self.n = pixel_count if pixel_count > 0 else w * h - pixel_begin
        self.pixel_begin = pixel_begin
        _check(lib().pt_init(C.byref(scene.desc), C.byref(opt)))
        self._live = True

    def render(self, iter_first: int, iter_count: int) -> None:
        _check(lib().pt_render(int(iter_first), int(iter_count)))

    def sync(self) -> None:
        _check(lib().pt_sync())

    def readback(self) -> np.ndarray:
        """Running SUM image of the tile, float32 [n, 3]."""
        out = np.empty((self.n, 3), np.float32)
        _check(lib().pt_readback(_f(out)))
        return out

    def readback_device(self, dev_ptr: int) -> None:
        _check(lib().pt_readback_device(C.c_void_p(dev_ptr)))

    def save_u8(self, samples: float) -> np.ndarray:
        """saveImage()'s bytes computed on the device: uint8 [rows, W, 3], x mirrored (whole-row tiles only)."""
        w, _ = self.scene.resolution
        out = np.empty((self.n // w, w, 3), np.uint8)
        _check(lib().pt_save_u8(C.c_float(samples), out.ctypes.data_as(C.POINTER(C.c_uint8))))
        return out

    def preview(self, iterations: int) -> np.ndarray:
        out = np.empty((self.n, 4), np.uint8)
        _check(lib().pt_preview_rgba8(int(iterations), out.ctypes.data_as(C.POINTER(C.c_uint8))))
        return out

    def stats(self) -> PtStats:
        st = PtStats()
        _check(lib().pt_get_stats(C.byref(st)))
        return st

    def reset_stats(self) -> None:
        _check(lib().pt_reset_stats())

    def clear(self) -> None:
        """Restart the accumulation (SUM image and statistics zeroed) on the same, already touched buffers."""
        _check(lib().pt_clear())

    def free(self) -> None:
        if self._live:
            self._live = False
            _check(lib().pt_free())

    # ---- stage-level entry points (parity tests) ----
    @staticmethod
    def stage_generate(pix_begin: int, n: int):
        o = np.zeros((3, n), np.float32)
        d = np.zeros((3, n), np.float32)
        _check(lib().pt_stage_generate(pix_begin, n, _f(o), _f(d)))
        return o, d

    @staticmethod
    def stage_intersect(o: np.ndarray, d: np.ndarray):
        n = o.shape[1]
        o = np.ascontiguousarray(o, np.float32)
        d = np.ascontiguousarray(d, np.float32)
        t = np.zeros(n, np.float32)
        nrm = np.zeros((3, n), np.float32)
        mat = np.zeros(n, np.int32)
        pt = np.zeros((3, n), np.float32)
        _check(lib().pt_stage_intersect(n, _f(o), _f(d), _f(t), _f(nrm), _i(mat), _f(pt)))
        return dict(t=t, nrm=nrm, mat=mat, pt=pt)

    @staticmethod
    def stage_save_u8(rgb_sum: np.ndarray, w: int, h: int, samples: float) -> np.ndarray:
        a = np.ascontiguousarray(rgb_sum, np.float32)
        out = np.empty((h, w, 3), np.uint8)
        _check(lib().pt_stage_save_u8(w, h, C.c_float(samples), _f(a), out.ctypes.data_as(C.POINTER(C.c_uint8))))
        return out

    @staticmethod
    def stage_shade(depth: int, it, pixel, hit: dict, o, d, color):
        n = o.shape[1]
        o, d, color = (np.ascontiguousarray(a, np.float32).copy() for a in (o, d, color))
        alive = np.zeros(n, np.int32)
        _check(lib().pt_stage_shade(n, depth, _i(np.ascontiguousarray(it, np.int32)),
                                    _i(np.ascontiguousarray(pixel, np.int32)), _f(hit["t"]), _f(hit["nrm"]),
                                    _i(hit["mat"]), _f(hit["pt"]), _f(o), _f(d), _f(color), _i(alive)))
        return o, d, color, alive


def pt_free() -> None:
    _check(lib().pt_free())


TRANSPORT = {"auto": 0, "rccl": 1, "copy": 2}  # PT_GROUP_TRANSPORT_* (include/pt_amd.h)


class Group:
    """pt_group_*: one process driving several GPUs, row-interleaved tiles, one exchange at write-out (RCCL; peer /
    device copies when `devices` names a device more than once — several contexts on one GPU — or on request)."""

    def __init__(self, scene: Scene, devices: Sequence[int], transport: str = "auto", **kw):
        self.scene = scene
        self._h = C.c_void_p()
        dev = np.asarray(list(devices), np.int32)
        opt = make_options(**kw)
        _check(lib().pt_group_create_ex(C.byref(scene.desc), C.byref(opt), _i(dev), len(dev), TRANSPORT[transport],
                                        C.byref(self._h)))

    @property
    def transport(self) -> str:
        t = lib().pt_group_transport(self._h)
        return {v: k for k, v in TRANSPORT.items()}[t]

    def render(self, iter_first: int, iter_count: int) -> None:
        _check(lib().pt_group_render(self._h, int(iter_first), int(iter_count)))

    def gather(self) -> np.ndarray:
        w, h = self.scene.resolution
        out = np.empty((w * h, 3), np.float32)
        _check(lib().pt_group_gather(self._h, _f(out)))
        return out

    def gather_u8(self, samples: float) -> np.ndarray:
        w, h = self.scene.resolution
        out = np.empty((h, w, 3), np.uint8)
        _check(lib().pt_group_gather_u8(self._h, C.c_float(samples), out.ctypes.data_as(C.POINTER(C.c_uint8))))
        return out

    def preview(self, iterations: int) -> np.ndarray:
        """Progressive preview of the running average (sendImageToPBO on every device + one exchange): uint8 [H*W, 4]."""
        w, h = self.scene.resolution
        out = np.empty((w * h, 4), np.uint8)
        _check(lib().pt_group_preview_rgba8(self._h, int(iterations), out.ctypes.data_as(C.POINTER(C.c_uint8))))
        return out

    def stats(self, i: int = 0) -> PtStats:
        st = PtStats()
        _check(lib().pt_ctx_get_stats(lib().pt_group_context(self._h, i), C.byref(st)))
        return st

    def free(self) -> None:
        if self._h:
            lib().pt_group_destroy(self._h)
            self._h = C.c_void_p()


def write_png_rgb8(path: str, rgb8: np.ndarray) -> None:
    a = np.ascontiguousarray(rgb8, np.uint8)
    h, w = a.shape[0], a.shape[1]
    if lib().pt_write_png_rgb8(os.fsencode(path), a.ctypes.data_as(C.POINTER(C.c_uint8)), w, h) != 0:
        raise PtError(f"cannot write {path}")


def output_basename(name: str, samples: int) -> str:
    buf = C.create_string_buffer(512)
    lib().pt_output_basename(name.encode(), int(samples), buf, 512)
    return buf.value.decode()


def save_png(path: str, rgb_sum: np.ndarray, w: int, h: int, samples: float) -> None:
    a = np.ascontiguousarray(rgb_sum, np.float32)
    if lib().pt_save_png(os.fsencode(path), _f(a), w, h, C.c_float(samples)) != 0:
        raise PtError(f"cannot write {path}")


def save_hdr(path: str, rgb_sum: np.ndarray, w: int, h: int, samples: float) -> None:
    """image::saveHDR's Radiance file (x mirrored, sum / samples), byte for byte the reference writer's."""
    a = np.ascontiguousarray(rgb_sum, np.float32)
    if lib().pt_save_hdr(os.fsencode(path), _f(a), w, h, C.c_float(samples)) != 0:
        raise PtError(f"cannot write {path}")


def save_pfm(path: str, rgb_sum: np.ndarray, w: int, h: int, samples: float) -> None:
    a = np.ascontiguousarray(rgb_sum, np.float32)
    if lib().pt_save_pfm(os.fsencode(path), _f(a), w, h, C.c_float(samples)) != 0:
        raise PtError(f"cannot write {path}")
