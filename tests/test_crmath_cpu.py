"""pydegensac_amd/csrc/dg_crmath.h compiled for the host: pow(x, 1.0/3), acos and cos as the device's rroots3 (Ftools.c:251-298)
takes them must be the CORRECTLY ROUNDED values — the host libm's in all but its own rare misroundings — so that the 7-point
solver's models carry the reference's bits.  Checked against 200-bit arithmetic (mpmath)."""
import ctypes as C, math, os, subprocess, tempfile
import numpy as np
import pytest

mp = pytest.importorskip("mpmath")
HDR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "pydegensac_amd", "csrc", "dg_crmath.h")


@pytest.fixture(scope="module")
def crlib():
    d = tempfile.mkdtemp()
    src = os.path.join(d, "h.c")
    with open(src, "w") as f:
        f.write(f'#include "{HDR}"\n'
                "void v_cos(const double *x, int n, double *o) { for (int i = 0; i < n; i++) o[i] = dg_cr_cos(x[i]); }\n"
                "void v_acos(const double *x, int n, double *o) { for (int i = 0; i < n; i++) o[i] = dg_cr_acos(x[i]); }\n"
                "void v_pow13(const double *x, int n, double *o) { for (int i = 0; i < n; i++) o[i] = dg_cr_pow13(x[i]); }\n")
    so = os.path.join(d, "h.so")
    subprocess.check_call(["gcc", "-O2", "-ffp-contract=off", "-fPIC", "-shared", "-o", so, src, "-lm"])
    return C.CDLL(so)


def _misrounded(fn, xs, exact, libm):
    xs = np.ascontiguousarray(xs, dtype=np.float64); out = np.zeros_like(xs)
    fn(xs.ctypes.data_as(C.POINTER(C.c_double)), len(xs), out.ctypes.data_as(C.POINTER(C.c_double)))
    mp.mp.prec = 200
    want = np.array([float(exact(mp.mpf(float(x)))) for x in xs])
    lib = np.array([libm(float(x)) for x in xs])
    return int((out != want).sum()), int((lib != want).sum())


def test_cos_on_the_solver_range_is_correctly_rounded(crlib):
    rng = np.random.default_rng(1); N = 12000
    xs = np.concatenate([rng.uniform(0, 2.1, N), rng.uniform(0, math.pi, N // 4), math.pi / 2 + rng.normal(0, 1e-6, N // 10),
                         [0.0, math.pi, math.pi / 2, math.pi / 4, 3 * math.pi / 4, -8.9e-16]])
    bad, bad_lib = _misrounded(crlib.v_cos, xs, mp.cos, math.cos)
    assert bad == 0, (bad, bad_lib)
    assert bad_lib < 0.01 * len(xs)                 # the premise: the host's libm is correctly rounded almost always


def test_acos_is_correctly_rounded_also_next_to_plus_and_minus_one(crlib):
    rng = np.random.default_rng(2); N = 12000
    xs = np.concatenate([rng.uniform(-1, 1, N), 1 - np.exp(rng.uniform(-35, 0, N // 4)), -1 + np.exp(rng.uniform(-35, 0, N // 4)),
                         rng.normal(0, 1e-8, N // 10), [0.5, -0.5, 0.0, 1.0, -1.0]])
    bad, bad_lib = _misrounded(crlib.v_acos, xs, mp.acos, math.acos)
    assert bad == 0, (bad, bad_lib)
    assert bad_lib < 0.01 * len(xs)


def test_pow_one_third_as_a_double_is_correctly_rounded(crlib):
    rng = np.random.default_rng(3); N = 12000
    third = None
    def exact(x):
        return mp.power(x, mp.mpf(1.0 / 3))         # the DOUBLE 1.0/3 = 1/3 - 2^-54/3, as the reference passes it
    xs = np.concatenate([np.exp(rng.uniform(-600, 600, N)), np.exp(rng.uniform(-3, 3, N)), [1.0, 8.0, 27.0, 1e-300, 1e300]])
    bad, bad_lib = _misrounded(crlib.v_pow13, xs, exact, lambda x: math.pow(x, 1.0 / 3))
    assert bad == 0, (bad, bad_lib)
    assert bad_lib < 0.01 * len(xs)
