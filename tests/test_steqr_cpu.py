"""CPU: the dsteqr of pydegensac_amd/csrc/dg_steqr9.h (the rotation chain of the wave eigen-solver),
compiled for the host, against the CPU oracle's dsyev restatement (oracle/dg_small.h dg_eig_sym, itself pinned on the
reference's LAPACK results by the golden tests) on tridiagonal inputs, where dsytd2 / dorg2l are the identity and
dsyev == dsteqr + ordering: eigenvalues and eigenvectors bit for bit, over QL and QR blocks, split matrices (exact and
negligible zeros on the subdiagonal), graded and clustered spectra."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))


@pytest.fixture(scope="module")
def host(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("steqr") / "libsteqr_host.so")
    subprocess.check_call(["g++", "-O2", "-ffp-contract=off", "-fPIC", "-shared", "-w", os.path.join(HERE, "steqr_host.cpp"), "-o", out])
    return C.CDLL(out)


def _cases(rng, count):
    for t in range(count):
        d = rng.normal(size=9); e = rng.normal(size=8)
        k = t % 10
        if k == 1:
            d *= 10.0 ** rng.integers(-6, 7); e *= 10.0 ** rng.integers(-6, 7)
        elif k == 2:
            e[rng.integers(0, 8)] = 0.0                                   # exact split
        elif k == 3:
            e[rng.integers(0, 8)] *= 1e-20                                # negligible subdiagonal
        elif k == 4:
            d = np.sort(np.abs(d))[::-1] * 10.0 ** np.arange(0, -9, -1)   # graded downwards
        elif k == 5:
            d = np.sort(np.abs(d)) * 10.0 ** np.arange(-8, 1)             # graded upwards -> the other direction
        elif k == 6:
            d[:] = 1.0 + 1e-9 * rng.normal(size=9); e *= 1e-6             # clustered
        elif k == 7:
            # what the estimator feeds it: the tridiagonal form of a 9x9 Gram matrix of normalised correspondences
            m = rng.normal(size=(rng.integers(8, 40), 9)); a = m.T @ m
            import scipy.linalg as sl
            h = sl.hessenberg(a); d = np.diag(h).copy(); e = np.diag(h, 1).copy()
        elif k == 8:
            e[[2, 5]] = 0.0; d[3:5] *= 1e5
        elif k == 9:
            d[:] = 0.0
        yield np.ascontiguousarray(d), np.ascontiguousarray(e)


def test_steqr9_matches_oracle_bitwise(host):
    rng = np.random.default_rng(5)
    n_ql = 0
    for d, e in _cases(rng, 20000):
        w0 = np.zeros(9); v0 = np.zeros(81); w1 = np.zeros(9); v1 = np.zeros(81)
        i0 = host.t_oracle_tridiag(dp(d), dp(e), dp(w0), dp(v0))
        i1 = host.t_steqr9(dp(d), dp(e), dp(w1), dp(v1))
        assert i1 != -99, "the recurrence differed between rows"
        assert i0 == i1
        assert w0.tobytes() == w1.tobytes(), (d, e, w0, w1)
        # dorg2l leaves -0.0 in the oracle's starting identity (-tau * 0 with tau = 0): zeros may differ in sign, nothing else
        assert np.array_equal(v0, v1), (d, e)
