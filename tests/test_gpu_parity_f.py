"""GPU parity of the fundamental-matrix path: the HIP kernel (through the C-ABI) against the CPU
oracle on the same seeded inputs.  Integer outputs (mask, sample / LO / model counts) bit-exact;
F within 1e-6 relative Frobenius (north_star tolerance)."""
import numpy as np
import pytest

import pydegensac_amd as pd
from pydegensac_amd import synthetic as syn

pytestmark = pytest.mark.gpu


def relF(a, b):
    a = a / np.linalg.norm(a); b = b / np.linalg.norm(b)
    return np.linalg.norm(a - b)


def run_pair(port, p1, p2, seed, **kw):
    kwo = dict(px_th=kw.get("px_th", 0.5), conf=kw.get("conf", 0.9999), max_iters=kw.get("max_iters", 100000),
               error_type=kw.get("error_type", 0), sym_check=kw.get("sym_check", True), degen=kw.get("degen", True),
               laf_coef=kw.get("laf_coef", 0.0))
    Fo, mo, so = port.find_fundamental(p1, p2, seed=seed, **kwo)
    F, m = pd.findFundamentalMatrix_(p1, p2, kwo["px_th"], kwo["conf"], kwo["max_iters"], kwo["error_type"],
                                     kwo["sym_check"], kwo["laf_coef"], kwo["degen"], seed=seed)
    st = pd.last_stats()
    return (F, m, st), (Fo, mo, so)


CASES = [
    ("c2", dict(n=2000, ir=0.4, sigma=0.1), {}),
    ("c2b_plane", dict(n=2000, ir=0.4, sigma=0.1, plane=0.7), dict(max_iters=3000)),
    ("n500", dict(n=500, ir=0.5, sigma=0.1), dict(max_iters=20000)),
    ("n100_low", dict(n=100, ir=0.3, sigma=0.3), dict(max_iters=5000)),
    ("n20", dict(n=20, ir=0.8, sigma=0.1), dict(max_iters=2000)),
    ("n8", dict(n=8, ir=1.0, sigma=0.1), dict(max_iters=200)),
    ("symm_epipolar", dict(n=1000, ir=0.4, sigma=0.1), dict(error_type=1)),
    ("nodegen", dict(n=1000, ir=0.4, sigma=0.1, plane=0.7), dict(degen=False)),
    ("nosym", dict(n=1000, ir=0.4, sigma=0.3), dict(sym_check=False)),
    ("noise_at_threshold", dict(n=1000, ir=0.4, sigma=0.5), dict(max_iters=8000)),
    ("all_outliers", dict(n=300, ir=0.0, sigma=0.5), dict(max_iters=3000)),
]


@pytest.mark.parametrize("name,gen,kw", CASES, ids=[c[0] for c in CASES])
@pytest.mark.parametrize("seed", [1, 7])
def test_fundamental_matches_oracle(oracle_port, name, gen, kw, seed):
    p1, p2, lab, _ = syn.two_view_fundamental(gen["n"], gen["ir"], gen["sigma"], seed=3, plane_fraction=gen.get("plane", 0.0))
    (F, m, st), (Fo, mo, so) = run_pair(oracle_port, p1, p2, seed, **kw)
    assert st["samples"] == so["samples"]
    assert st["lo_runs"] == so["lo_runs"]
    assert st["degen"] == so["degen"]
    assert st["full_passes"] == so["full_passes"] and st["ex_passes"] == so["ex_passes"]
    assert np.array_equal(np.asarray(m), mo), f"{(np.asarray(m) != mo).sum()} mask bits differ"
    if np.abs(Fo).sum() == 0:
        assert np.abs(F).sum() == 0
    else:
        assert relF(F, Fo) < 1e-6


def test_sample_stream_matches_glibc_replay(oracle_port):
    """the device sampler (seed chain + draws + pool swaps) against the oracle's libc-faithful replay"""
    import ctypes as C
    from pydegensac_amd import _lib
    L = _lib.lib()
    # n <= 4096 runs the parallel pool stage (pool in LDS), larger n the sequential one; tiny n = every draw aliases
    for ssz, n, iters in [(7, 2000, 700), (4, 5000, 600), (7, 9, 300), (4, 5, 300), (4, 3000, 900), (7, 64, 600), (7, 4096, 1000), (7, 8, 256)]:
        out = np.zeros((iters, ssz), np.int32)
        _lib.check(L.mi_degensac_sample_stream(12345, n, ssz, iters, 0, out.ctypes.data_as(C.POINTER(C.c_int32))))
        ref = np.zeros((iters, ssz), np.int32)
        oracle_port.lib().dg_oracle_sample_stream(C.c_uint(12345), n, ssz, iters, oracle_port.ip(ref), None)
        assert np.array_equal(out[:, ::-1], ref)        # the oracle lists samidx order = reverse draw order


def test_scoring_kernel_residuals_bit_exact(oracle_port):
    import ctypes as C
    from pydegensac_amd import _lib
    L = _lib.lib()
    p1, p2, lab, Fgt = syn.two_view_fundamental(3000, 0.4, 0.1, seed=0)
    n = 3000
    rng = np.random.default_rng(0)
    models = np.concatenate([Fgt.reshape(1, 9), rng.normal(size=(15, 9))]).copy()
    u = np.ones((n, 6)); u[:, 0:2] = p1; u[:, 3:5] = p2
    for kind, fn in [(0, "dg_oracle_FDs"), (1, "dg_oracle_FDsSym"), (10, "dg_oracle_HDs")]:
        I = np.zeros(16, np.uint32); J = np.zeros(16); res = np.zeros((16, n))
        _lib.check(L.mi_degensac_score_models(_lib.dptr(p1), _lib.dptr(p2), n, 2, _lib.dptr(models), 16, kind, 0.25, 0,
                                              I.ctypes.data_as(C.POINTER(C.c_uint32)), _lib.dptr(J), _lib.dptr(res)))
        for k in range(16):
            d = np.zeros(n)
            getattr(oracle_port.lib(), fn)(oracle_port.dp(u), oracle_port.dp(models[k].copy()), oracle_port.dp(d), n)
            inl = np.zeros(n, np.int32)
            S = oracle_port.lib().dg_oracle_inlidxs(oracle_port.dp(d), n, C.c_double(0.25), oracle_port.ip(inl))
            assert np.array_equal(d, res[k]), (kind, k)                 # residuals bit-exact (IEEE div/sqrt, no contraction)
            assert S.I == I[k] and S.J == J[k], (kind, k, S.J, J[k])      # J is the reference's sequential sum, bit for bit
