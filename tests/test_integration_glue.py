"""INTEGRATION.md §2 is code that compiles: tests/integration/pathtrace_amd_glue.cpp against the reference's own
headers (scene.h, pathtrace.h, sceneStructs.h, GLM, the genuine cuda_runtime.h for uchar4) with static_asserts on
every Material / Camera field offset; linked with the reference's loader into oracle/_ref/ref_glue_demo, which must
fail LOUDLY without a GPU (no CPU fallback behind the reference's entry points either).  Only where the reference is
mounted; the GPU-side run is tests/test_gpu_cli.py::test_reference_side_glue_end_to_end."""
import os
import shutil
import subprocess

import pytest

from conftest import has_gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"

pytestmark = pytest.mark.skipif(not (os.path.isdir(REF + "/src") and shutil.which("g++")), reason="reference not mounted")


def _cuda_inc():
    import importlib.util
    s = importlib.util.find_spec("triton")
    d = os.path.join(os.path.dirname(s.origin), "backends", "nvidia", "include") if s else ""
    return d if d and os.path.isfile(os.path.join(d, "cuda_runtime.h")) else None


def test_glue_compiles_against_reference_headers(tmp_path):
    inc = _cuda_inc()
    if not inc:
        pytest.skip("no <cuda_runtime.h> in this image")
    src = os.path.join(ROOT, "tests", "integration", "pathtrace_amd_glue.cpp")
    cmd = ["g++", "-std=c++14", "-Wall", "-Wno-unknown-pragmas", "-Wno-attributes", "-c", src, "-o", str(tmp_path / "glue.o"),
           f"-I{REF}/src", f"-I{REF}/external/include", f"-I{inc}", f"-I{ROOT}/include"]
    p = subprocess.run(cmd, capture_output=True, text=True)
    assert p.returncode == 0, p.stderr[-3000:]
    syms = subprocess.run(["nm", "-C", str(tmp_path / "glue.o")], capture_output=True, text=True).stdout
    for want in ("T pathtraceInit(Scene*)", "T pathtraceFree()", "T pathtrace(uchar4*, int, int)",
                 "T InitDataContainer(GuiDataContainer*)"):
        assert want in syms, want  # exactly the four signatures of src/pathtrace.h:6-9
    for need in ("U pt_init", "U pt_render", "U pt_readback", "U pt_free", "U pt_preview_rgba8_device", "U pt_last_error"):
        assert need in syms, need


def test_inline_glue_in_integration_md_is_the_compiled_file():
    """The code block of INTEGRATION.md §2 is an excerpt of the compiled file, not a second version of it."""
    md = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    assert "tests/integration/pathtrace_amd_glue.cpp" in md
    src = open(os.path.join(ROOT, "tests", "integration", "pathtrace_amd_glue.cpp")).read()
    block = md.split("```cpp", 1)[1].split("```", 1)[0]
    for line in block.splitlines():
        line = line.strip()
        if line and not line.startswith("//"):
            assert line in src, f"INTEGRATION.md shows a line the compiled glue does not contain: {line}"


@pytest.mark.skipif(has_gpu(), reason="the GPU run is in tests/test_gpu_cli.py")
def test_glue_demo_fails_loudly_without_gpu(tmp_path):
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "ref"], stdout=subprocess.DEVNULL,
                          stderr=subprocess.DEVNULL)
    exe = os.path.join(ROOT, "oracle", "_ref", "ref_glue_demo")
    if not os.path.exists(exe):
        pytest.skip("no <cuda_runtime.h> in this image")
    p = subprocess.run([exe, os.path.join(ROOT, "tests", "golden", "scenes", "ref_quirks.txt"), str(tmp_path / "o.f32")],
                       capture_output=True, text=True)
    assert p.returncode != 0 and "HIP error (pathtraceInit)" in p.stderr and not os.path.exists(tmp_path / "o.f32")
