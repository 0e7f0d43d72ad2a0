"""GPU: the per-LO residual dump (include/mi_degensac.h MI_DEGENSAC_RESIDS_M) against the dump the UNMODIFIED
reference fills and frees (oracle/_ref, captured by ref_shim.c).  Row 0 of the first run (a minimal-sample model) is
bit-exact; the other rows are residuals of least-squares models (also row 0 of a run started from the DEGENSAC branch), which agree with the reference's to the last bits of its LAPACK
build's dsyev (models within 1e-9), so they are compared to 3e-6 / sqrt(residual) relative, at most 2e-5 (see _check); rows the reference memsets keep
its byte pattern; asking for the dump changes nothing else."""
import numpy as np
import pytest

import pydegensac_amd as pd
from pydegensac_amd import diagnostics as dg, synthetic as syn

pytestmark = pytest.mark.gpu


def _check(res, ref, lo_runs, n):
    runs = min(lo_runs, res.shape[0])
    assert runs > 0
    written = ~np.isnan(res[:runs]).all(axis=2)                       # rows the kernel computed (or memset)
    assert written[:, 1].all()                                        # the LSQ-before-LO row exists in every run
    for r in range(runs):
        for row in np.flatnonzero(written[r]):
            a, b = res[r, row], ref[r, row]
            if (row == 0 and r == 0) or not np.isfinite(b).all():
                assert np.array_equal(a.view(np.uint64), b.view(np.uint64)), (r, row)
            else:
                # Least-squares models agree with the reference's to the last bits of its LAPACK build (<= 1e-9 relative,
                # DESIGN.md 6).  A residual is d = r^2 / den, and the model's 1e-9 moves r by a fixed absolute amount (r is a
                # sum of terms ~1e3 .. 1e6 times a well-fitting r), so the relative change of d falls like 1 / sqrt(d):
                # tolerance 3e-6 / sqrt(d) relative, capped at 2e-5 for residuals far below one squared pixel (3e-7 at
                # d = 100, 3e-8 at d = 1e4, never below 1e-8: the models themselves agree to 1e-9).  A row taken from the wrong buffer (another iterate of the same LO) differs by
                # 1e-3 .. 1 relative on most points and fails.
                rt = np.maximum(1e-8, np.minimum(2e-5, 3e-6 / np.sqrt(np.maximum(np.abs(b), 1e-300))))
                err = np.abs(a - b)
                bad = err > rt * np.abs(b) + 1e-12
                assert not bad.any(), (r, row, int(bad.sum()), float((err / np.maximum(np.abs(b), 1e-300))[bad].max()), float(np.abs(b)[bad].min()))
    assert np.isnan(res[runs:]).all()
    return int(written.sum())


@pytest.mark.parametrize("et", [0, 1])
def test_fundamental_residual_dump_equals_reference(oracle_ref, et):
    p1, p2, _, _ = syn.two_view_fundamental(1200, 0.4, 0.1, seed=3)
    F, m, st, res = dg.find_fundamental_with_residuals(p1, p2, 0.5, 0.9999, 20000, et, seed=5, lo_runs=12)
    ref, (Fo, mo, so) = oracle_ref.resids_of("F", p1, p2, 12, px_th=0.5, conf=0.9999, max_iters=20000, error_type=et, seed=5)
    assert st["lo_runs"] == so["lo_runs"] and np.array_equal(m, mo)
    rows = _check(res, ref, st["lo_runs"], 1200)
    assert rows > 20 * min(st["lo_runs"], 12)
    F2, m2 = pd.findFundamentalMatrix_(p1, p2, 0.5, 0.9999, 20000, et, seed=5)
    assert np.array_equal(F, F2) and np.array_equal(m, m2)           # asking for the dump changes nothing else


@pytest.mark.parametrize("et", [0, 2, 3])
def test_homography_residual_dump_equals_reference(oracle_ref, et):
    p1, p2, _, _ = syn.homography_pairs(900, 0.4, 0.5, seed=4)
    H, m, st, res = dg.find_homography_with_residuals(p1, p2, 2.0, 0.999, 20000, et, seed=7, lo_runs=6)
    ref, (Ho, mo, so) = oracle_ref.resids_of("H", p1, p2, 6, px_th=2.0, conf=0.999, max_iters=20000, error_type=et, seed=7)
    assert st["lo_runs"] == so["lo_runs"] and np.array_equal(m, mo)
    _check(res, ref, st["lo_runs"], 900)


def test_few_inliers_rows_keep_the_reference_memset():
    """fewer than 16 (F) inliers after the LSQ: the reference zero-fills the 60 repetition rows (exp_ranF.c:761)"""
    p1, p2, _, _ = syn.two_view_fundamental(60, 0.2, 0.1, seed=9)
    F, m, st, res = dg.find_fundamental_with_residuals(p1, p2, 0.5, 0.9999, 3000, 0, seed=2, lo_runs=4)
    if st["lo_runs"] > 0:
        r0 = res[0]
        assert not np.isnan(r0[1]).any()
        assert (r0[2:] == 0).all() or not np.isnan(r0[2]).all()


@pytest.mark.parametrize("n,ir,mi", [(800, 0.4, 6000), (300, 0.6, 2000), (1500, 0.3, 12000)])
def test_support_histogram_equals_reference_data_out(oracle_ref, n, ir, mi):
    """the F driver's data_out histogram (exp_ranF.c:1495): exact, incl. the sample / LO counters in front"""
    p1, p2, _, _ = syn.two_view_fundamental(n, ir, 0.1, seed=6, plane_fraction=0.5 if n == 1500 else 0.0)
    F, m, st, hist = dg.find_fundamental_with_support_histogram(p1, p2, 0.5, 0.9999, mi, seed=3)
    ref, (Fo, mo, so) = oracle_ref.data_out_of(p1, p2, px_th=0.5, conf=0.9999, max_iters=mi, seed=3)
    assert np.array_equal(hist, ref), np.flatnonzero(hist != ref)[:10]
    assert hist[0] == so["samples"] and hist[1] == so["lo_runs"] and np.array_equal(m, mo)
    F2, m2 = pd.findFundamentalMatrix_(p1, p2, 0.5, 0.9999, mi, seed=3)
    assert np.array_equal(F, F2) and np.array_equal(m, m2)           # switching the screens off changes nothing
