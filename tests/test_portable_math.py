"""pt_portable_math.h (shared by the HIP kernels and the oracle's PORTABLE mode) against libm.  CPU only."""
import numpy as np


def ulp_diff(a, b):
    ia = a.view(np.int32).astype(np.int64)
    ib = b.view(np.int32).astype(np.int64)
    ia = np.where(ia < 0, -(ia & 0x7FFFFFFF), ia)
    ib = np.where(ib < 0, -(ib & 0x7FFFFFFF), ib)
    return np.abs(ia - ib)


def test_float_functions_accuracy(oracle):
    rng = np.random.default_rng(7)
    u = np.concatenate([rng.random(400000, dtype=np.float32), np.array([0.0, 1.0, 0.5, 2.0 ** -24], np.float32)])
    ang = (2.0 * np.pi * u.astype(np.float64)).astype(np.float32)  # [0, 2pi]: the range the renderer uses
    oracle.set_math_mode(oracle.PORTABLE)
    for fn, x, ref in ((0, ang, np.sin), (1, ang, np.cos), (2, u, np.arccos)):
        got = oracle.math_eval_f(fn, x)
        exact = ref(x.astype(np.float64)).astype(np.float32)  # correctly rounded (double libm, then one rounding)
        if fn < 2:  # sin/cos: double reduction + float kernels: <= 1 ulp (cos: 2 ulp for ~1e-5 of the arguments)
            err = ulp_diff(got, exact)
            assert err.max() <= (1 if fn == 0 else 2) and (err <= 1).mean() > 0.9999 and (err == 0).mean() > 0.7, fn
        else:       # acos: fdlibm float algorithm, < 1 ulp
            assert ulp_diff(got, exact).max() <= 1
        oracle.set_math_mode(oracle.LIBM)
        libm = oracle.math_eval_f(fn, x)
        oracle.set_math_mode(oracle.PORTABLE)
        assert ulp_diff(got, libm).max() <= (3 if fn == 1 else 2), fn  # same accuracy class as glibc's float routines
        if fn == 2:  # and acos reproduces glibc's acosf exactly
            assert np.array_equal(got.view(np.uint32), libm.view(np.uint32))


def test_double_functions(oracle):
    x = np.linspace(0.0, 2.0 * np.pi, 200001)
    oracle.set_math_mode(oracle.PORTABLE)
    # double ARGUMENTS (the specular branch), float-accurate results
    assert np.abs(oracle.math_eval_d(0, x) - np.sin(x)).max() < 1.5e-7
    assert np.abs(oracle.math_eval_d(1, x) - np.cos(x)).max() < 1.5e-7


def test_special_values(oracle):
    oracle.set_math_mode(oracle.PORTABLE)
    assert oracle.math_eval_f(2, np.array([1.0], np.float32))[0] == 0.0
    assert oracle.math_eval_f(2, np.array([0.0], np.float32))[0] == np.float32(np.pi / 2)
    assert np.isnan(oracle.math_eval_f(2, np.array([1.5], np.float32))[0])
    assert oracle.math_eval_f(0, np.array([0.0], np.float32))[0] == 0.0
    assert oracle.math_eval_f(1, np.array([0.0], np.float32))[0] == 1.0
