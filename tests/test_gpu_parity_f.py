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


# ---- findFundamentalMatrix with [N, 6] input and laf_consistensy_coef > 0 (utils.py:111-146; exp_ranF.c:1394-1411, :1536-1556,
# :1664-1682, final filter :1724-1739; LAF point sets bindings.cpp:337-409) ----
LAF_CASES = [
    ("laf_c2", dict(n=2000, ir=0.4, sigma=0.1), dict(laf_coef=3.0)),
    ("laf_c2_symm_epipolar", dict(n=2000, ir=0.4, sigma=0.1), dict(laf_coef=3.0, error_type=1)),
    ("laf_plane", dict(n=2000, ir=0.4, sigma=0.1, plane=0.7), dict(laf_coef=2.0, max_iters=20000)),
    ("laf_half_bad_tight", dict(n=1000, ir=0.5, sigma=0.3, bad=0.5), dict(laf_coef=1.0, px_th=1.0, max_iters=20000)),
    ("laf_nosym_n300", dict(n=300, ir=0.6, sigma=0.3, bad=0.5, lsig=0.5), dict(laf_coef=1.0, sym_check=False, max_iters=5000)),
]


def _laf_pair(gen):
    return syn.two_view_fundamental(gen["n"], gen["ir"], gen["sigma"], seed=3, plane_fraction=gen.get("plane", 0.0), laf=True,
                                    laf_bad=gen.get("bad", 0.25), laf_sigma=gen.get("lsig", 0.05))[:2]


@pytest.mark.parametrize("name,gen,kw", LAF_CASES, ids=[c[0] for c in LAF_CASES])
@pytest.mark.parametrize("variant", [1, 2, 3], ids=["t512", "t256", "t128"])
@pytest.mark.parametrize("placement", [1, 2, 3], ids=["hbm", "lds", "pool_lds"])
def test_fundamental_with_laf_check_matches_oracle(oracle_port, name, gen, kw, variant, placement):
    """every workgroup size x placement; the final LAF filter (MI_DEGENSAC_FLAG_FINAL_LAF_FILTER) off and on"""
    p1, p2 = _laf_pair(gen)
    assert p1.shape[1] == 6
    tn = variant | (placement << 2)
    for seed in (1, 7):
        for fin in (0, 1):
            kwo = dict(px_th=kw.get("px_th", 0.5), conf=0.9999, max_iters=kw.get("max_iters", 100000), error_type=kw.get("error_type", 0),
                       sym_check=kw.get("sym_check", True), degen=True, laf_coef=kw["laf_coef"])
            Fo, mo, so = oracle_port.find_fundamental(p1, p2, seed=seed, final_laf_filter=bool(fin), **kwo)
            F, m = pd.findFundamentalMatrix_(p1, p2, kwo["px_th"], kwo["conf"], kwo["max_iters"], kwo["error_type"], kwo["sym_check"],
                                             kwo["laf_coef"], True, seed=seed, flags=fin, tuning=tn)
            st = pd.last_stats()
            assert (st["samples"], st["lo_runs"], st["degen"], st["rejected"]) == (so["samples"], so["lo_runs"], so["degen"], so["rejected"]), (seed, fin)
            assert st["full_passes"] == so["full_passes"] and st["ex_passes"] == so["ex_passes"]
            assert np.array_equal(np.asarray(m), mo), f"{(np.asarray(m) != mo).sum()} mask bits differ (seed {seed}, final filter {fin})"
            assert relF(F, Fo) < 1e-6


def test_laf_check_rejects_candidates_and_changes_the_run(oracle_port):
    """the check bites: over the LAF cases candidates ARE turned down on `S.Ilafs < maxS.Ilafs` (counter MI_ST_REJECTED, equal on both
    sides), the final filter clears mask entries, and at least one run differs from the same run without the check"""
    rej = 0; changed = 0; filtered = 0
    for name, gen, kw in LAF_CASES:
        p1, p2 = _laf_pair(gen)
        for seed in (1, 7, 11):
            a = (kw.get("px_th", 0.5), 0.9999, kw.get("max_iters", 100000), kw.get("error_type", 0), kw.get("sym_check", True))
            F, m = pd.findFundamentalMatrix_(p1, p2, *a, kw["laf_coef"], True, seed=seed); st = pd.last_stats()
            Fo, mo, so = oracle_port.find_fundamental(p1, p2, *a, kw["laf_coef"], True, seed=seed)
            assert st["rejected"] == so["rejected"] and st["samples"] == so["samples"] and np.array_equal(np.asarray(m), mo)
            rej += st["rejected"]
            F0, m0 = pd.findFundamentalMatrix_(p1, p2, *a, 0.0, True, seed=seed); st0 = pd.last_stats()
            assert st0["rejected"] == 0
            changed += (st0["samples"], st0["lo_runs"]) != (st["samples"], st["lo_runs"]) or not np.array_equal(np.asarray(m0), np.asarray(m))
            F1, m1 = pd.findFundamentalMatrix_(p1, p2, *a, kw["laf_coef"], True, seed=seed, flags=1)
            filtered += int(np.asarray(m).sum() - np.asarray(m1).sum())
            assert not (np.asarray(m1) & ~np.asarray(m)).any()          # the filter only clears entries
    assert rej > 0 and changed > 0 and filtered > 0, (rej, changed, filtered)


def test_public_api_laf_rows(oracle_port):
    """the drop-in call itself: findFundamentalMatrix(pts1[N,6], pts2[N,6], laf_consistensy_coef=3) (utils.py:111-146)"""
    p1, p2 = _laf_pair(dict(n=1000, ir=0.5, sigma=0.1))
    F, m = pd.findFundamentalMatrix(p1, p2, 0.5, 0.9999, 20000, 3.0, seed=5)
    Fo, mo, so = oracle_port.find_fundamental(p1, p2, 0.5, 0.9999, 20000, 0, True, 3.0, True, seed=5)
    assert np.array_equal(np.asarray(m, bool), mo) and relF(np.asarray(F), Fo) < 1e-6
