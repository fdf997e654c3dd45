"""MI355X-native wavefront path tracer — Python-side plumbing.

The product is the HIP library (csrc/ → libpt_amd.so, C ABI in include/pt_amd.h) and its C++
host (pt_scene.cpp, pathtrace_shim.cpp, pt_render).  This package only binds the C ABI for the
benchmark / test drivers (capi), synthesises scene files in the reference's text format
(scenes) and holds the multi-GPU tile partition + gather (parallel).
"""
__all__ = ["capi", "scenes", "parallel"]
