"""The exact and fma kernels (and depth 0 of every mode) take correctly rounded square roots, reciprocals and quotients from
instruction sequences shorter than the compiler's general expansions whenever every lane's operands are in a range where both
are the same computation (csrc/pt_kernels.hip, namespace ieee).  Bit-exactness against the oracle rests on that claim, so it is
checked ON THE DEVICE against the compiler's own `__builtin_sqrtf`, `1.0f / x` and `a / b`: exhaustively for the one-operand
forms (all 2^32 bit patterns: zeros, denormals, infinities and NaNs take the expansion's path, the rest the short one) and on
2^33 pseudo-random operand sets for the quotients (exponents inside the guarded range for 7 waves of 8, unrestricted for the
eighth; range edges oversampled; signed-zero numerators, which the short chain would get wrong (-0 / b -> +0) and the guard
therefore excludes)."""
import pytest

from cosc_4397_pathtracing_raytracing_project_amd import capi

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("arith", ["exact", "fma", "fast"])
@pytest.mark.parametrize("kind,name", [(0, "sqrt"), (1, "reciprocal"), (2, "reciprocal square root")])
def test_one_operand_forms_equal_the_compilers_for_every_float(arith, kind, name):
    assert capi.selfcheck_ieee(kind, 0, 1 << 32, arith=arith) == 0, name


@pytest.mark.parametrize("arith", ["exact", "fma", "fast"])
@pytest.mark.parametrize("kind,name", [(3, "a / b"), (4, "div2 / div3 / rcp3")])
def test_quotients_equal_the_compilers_on_random_operands(arith, kind, name):
    for seed in (1, 2):
        assert capi.selfcheck_ieee(kind, seed << 34, 1 << 32, seed=seed, arith=arith) == 0, (name, seed)
