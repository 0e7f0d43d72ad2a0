"""CPU: the lane-level small dense routines of pydegensac_amd/csrc/dg_mat3.h (3x3 inverse, 3x3 right singular
vectors, Hdetect, 9-column null space), compiled for the host, against the UNMODIFIED reference's own routines in
oracle/_ref (matutls/minv.c, matutls/svduv.c, DegUtils.c:84-161 Hdetect, utools.c:97-167 nullspace): bit for bit on
10^4 random inputs each, including rank-deficient and badly scaled ones."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))


@pytest.fixture(scope="module")
def host(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("mat3") / "libmat3_host.so")
    subprocess.check_call(["g++", "-O2", "-ffp-contract=off", "-fPIC", "-shared", "-w", os.path.join(HERE, "mat3_host.cpp"), "-o", out])
    return C.CDLL(out)


@pytest.fixture(scope="module")
def ref():
    from oracle import ref as R
    if not R.available():
        pytest.skip("oracle/_ref is not built (needs /root/reference)")
    return R.lib()


def _mats(rng, count):
    """random 3x3 inputs: generic, scaled over many decades, rank 2, rank 1, with exact zeros"""
    for t in range(count):
        a = rng.normal(size=(3, 3)) * 10.0 ** rng.integers(-3, 4)
        k = t % 8
        if k == 1:
            a[2] = a[0] * rng.normal() + a[1] * rng.normal()           # rank 2 up to rounding
        elif k == 2:
            u, s, vt = np.linalg.svd(a); s[2] = 0; a = (u * s) @ vt     # fundamental-matrix like
        elif k == 3:
            a = np.outer(rng.normal(size=3), rng.normal(size=3))        # rank 1
        elif k == 4:
            a[rng.integers(0, 3), rng.integers(0, 3)] = 0.0
        elif k == 5:
            a[:, 0] *= 1e-9
        elif k == 6:
            a[1] = a[0]                                                 # exactly repeated row
        yield np.ascontiguousarray(a)


def test_inv3_equals_reference_minv(host, ref):
    rng = np.random.default_rng(11); n_sing = 0
    for a in _mats(rng, 10000):
        x = a.copy(); y = a.copy()
        r1 = host.t_inv3(dp(x)); r2 = ref.minv(dp(y), 3)
        assert (r1 != 0) == (r2 != 0)
        assert np.array_equal(x, y, equal_nan=True), a          # also the partly factored matrix of a singular input
        n_sing += r1 != 0
    assert 0 < n_sing < 5000


def test_svd3_right_equals_reference_svduv(host, ref):
    rng = np.random.default_rng(12)
    for a in _mats(rng, 10000):
        x = a.copy(); v = np.zeros(9); d = np.zeros(3)
        host.t_svd3_right(dp(x), dp(v), dp(d))
        y = a.copy(); d2 = np.zeros(3); u2 = np.zeros(9); v2 = np.zeros(9)
        ref.svduv(dp(d2), dp(y), dp(u2), 3, dp(v2), 3)
        assert np.array_equal(d, d2, equal_nan=True) and np.array_equal(v, v2, equal_nan=True), a


def test_hdetect_equals_reference(host, ref):
    rng = np.random.default_rng(13)
    trip = np.array([[0, 1, 2], [3, 4, 5], [0, 1, 6], [3, 4, 6], [2, 5, 6]], np.uint8)
    for t in range(10000):
        F = rng.normal(size=(3, 3))
        if t % 3:
            u, s, vt = np.linalg.svd(F); s[2] = 0; F = (u * s) @ vt
        F = np.ascontiguousarray(F)
        pts = rng.uniform(-500, 500, size=(7, 4))
        if t % 7 == 0:
            pts[2] = pts[0] + (pts[1] - pts[0]) * 0.3                   # collinear triple: singular 3x3 system
        u7 = np.ones((7, 6)); u7[:, 0:2] = pts[:, 0:2]; u7[:, 3:5] = pts[:, 2:4]
        ids = np.ascontiguousarray(trip[t % 5])
        H1 = np.zeros(9); H2 = np.zeros(9)
        host.t_hdetect(dp(F), dp(np.ascontiguousarray(pts)), ids.ctypes.data_as(C.POINTER(C.c_ubyte)), dp(H1))
        ref.Hdetect(dp(F.copy()), dp(u7), ids.ctypes.data_as(C.POINTER(C.c_ubyte)), dp(H2))
        assert np.array_equal(H1, H2, equal_nan=True), (t, H1, H2)


def test_null9_equals_reference_nullspace(host, ref):
    rng = np.random.default_rng(14)
    seen = set()
    for t in range(10000):
        rows = (7, 8, 9)[t % 3]
        M = rng.normal(size=(rows, 9))
        k = (t // 3) % 6
        if k == 1: M[:, rng.integers(0, 9)] = 0.0                       # a column without a pivot
        if k == 2: M[rows - 1] = M[0]                                    # rank deficiency
        if k == 3: M[:, 1] = M[:, 0] * 2.0
        if k == 4: M[:, 0] *= 1e-13                                      # below the 1e-12 tolerance
        if k == 5 and rows == 9: M[8] = 0.0
        A = np.zeros((9, 9)); A[:rows] = M
        ns2 = np.zeros(81); buf = np.zeros(18, np.int32)
        n2 = ref.nullspace(dp(A.copy().ravel()), dp(ns2), 9, buf.ctypes.data_as(C.POINTER(C.c_int)))
        ns1 = np.zeros(18)
        n1 = host.t_null9(rows, dp(np.ascontiguousarray(M).ravel().copy()), dp(ns1))
        assert n1 == n2, (t, n1, n2)
        m = min(n1, 2)
        assert np.array_equal(ns1[:9 * m], ns2[:9 * m], equal_nan=True), t
        seen.add(n1)
    assert {1, 2, 3} <= seen
