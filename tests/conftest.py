import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _native_built():
    """Build the oracle (g++) and make sure the HIP library exists.  On the GPU box the
    .so files travel with the snapshot, so this is a no-op there unless sources changed."""
    from oracle import binding as ob
    ob.build()
    # always `make` (a no-op when up to date) rather than "build if missing": a stale or foreign libpt_amd.so must not
    # be what the tests measure.  On the GPU box hipcc is present too; if it is not, the travelled .so is used as is.
    import shutil
    import subprocess
    pkg = os.path.join(ROOT, "cosc_4397_pathtracing_raytracing_project_amd")
    if os.environ.get("PT_AMD_LIB"):
        return  # an explicitly chosen A/B build (tools/): leave it alone
    if shutil.which("make") and os.path.exists(os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")):
        subprocess.check_call(["make", "-C", os.path.join(pkg, "csrc"), "-j8", "all"], stdout=subprocess.DEVNULL)
    elif not os.path.exists(os.path.join(pkg, "libpt_amd.so")):
        import __graft_entry__ as ge
        ge.build()


@pytest.fixture(scope="session")
def scene_dir(tmp_path_factory):
    from cosc_4397_pathtracing_raytracing_project_amd import scenes
    d = tmp_path_factory.mktemp("scenes")
    paths = {
        "cornell": scenes.write_scene(scenes.cornell_scene_text(), str(d / "cornell.txt")),
        "sphere": scenes.write_scene(scenes.sphere_scene_text(), str(d / "sphere.txt")),
        "stress": scenes.write_scene(scenes.stress_scene_text((6, 5, 4), res=(160, 90)), str(d / "stress.txt")),
        # > 64 KB of BVH + geometry tables: exercises the kernels' global-memory table path and real subtrees
        "stress_big": scenes.write_scene(scenes.stress_scene_text((10, 10, 8), res=(160, 90)), str(d / "stress_big.txt")),
    }
    return paths


@pytest.fixture()
def oracle():
    from oracle import binding as ob
    ob.set_math_mode(ob.LIBM)
    yield ob
    ob.set_math_mode(ob.LIBM)


def has_gpu() -> bool:
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False
