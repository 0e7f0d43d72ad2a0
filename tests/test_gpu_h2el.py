"""GPU: ransacH2el (degensac/ranH2el.c:19) — RANSAC on ellipse-to-ellipse correspondences, two per sample —
against fixtures from the unmodified reference (tests/golden/E_*.npz) and against the CPU oracle on seeded problems."""
import os

import numpy as np
import pytest

import pydegensac_amd as pd
from pydegensac_amd import synthetic as syn
from pydegensac_amd import _lib
from tests import golden_util as gu

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("path", gu.fixtures("E"), ids=lambda p: os.path.basename(p)[:-4])
def test_ellipse_ransac_matches_reference_goldens(path):
    g = gu.load(path); kw = g["call"]
    H, m = pd.ransacH2el(g["p1"], kw["th"], kw["conf"], kw["max_iters"], kw.get("do_lo", True), kw.get("inl_limit", 0), seed=g["seed"], raw=True)
    st = pd.last_stats()
    assert (st["samples"], st["lo_runs"], st["I"]) == (g["samples"], g["lo_runs"], g["I"])
    assert np.array_equal(np.asarray(m), g["mask"])
    assert gu.rel(H, g["model"]) < 1e-6


def test_ellipse_ransac_ragged_batch_against_oracle(oracle_port):
    """a ragged batch (9 .. 4000 correspondences, 8-50 % inliers, LO with and without a fit limit): every pair against the
    oracle — sample / LO / scored-model counters, mask bit for bit, model to 1e-6"""
    cases = [(1000, 0.3, 1.0, 0.05), (4000, 0.08, 1.0, 0.05), (400, 0.15, 1.5, 0.1), (60, 0.5, 1.0, 0.05), (2500, 0.4, 0.5, 0.02),
             (30, 0.6, 0.5, 0.02), (9, 1.0, 0.2, 0.01), (700, 0.0, 1.0, 0.05)]
    U = [syn.ellipse_pairs(n, ir, sig, 900 + i, ln)[0] for i, (n, ir, sig, ln) in enumerate(cases)]
    seeds = [3 + 2 * i for i in range(len(U))]
    for do_lo, lim, th, mi in [(True, 0, 4.0, 10000), (True, 25, 9.0, 10000), (False, 0, 4.0, 2000)]:
        H, m = pd.ransacH2el_batch(U, th, 0.99, mi, do_lo, lim, seeds=seeds, raw=True)
        st = pd.last_stats()
        for p in range(len(U)):
            Ho, mo, so = oracle_port.ransacH2el(U[p], th, 0.99, mi, do_lo, lim, seeds[p])
            assert (st[p]["samples"], st[p]["lo_runs"], st[p]["I"], st[p]["models"]) == (so["samples"], so["lo_runs"], so["I"], so["models"]), (do_lo, lim, p)
            assert np.array_equal(np.asarray(m[p]), mo), (do_lo, lim, p)
            assert gu.rel(H[p], Ho) < 1e-6, (do_lo, lim, p)
        if do_lo:
            assert any(s_["lo_runs"] >= 1 for s_ in st), "no pair of the batch ran a local optimisation"
    # the default return value follows findHomography's convention: the conventional image 1 -> image 2 matrix = inv(raw.T)
    Hc, _ = pd.ransacH2el_batch(U[:2], 4.0, 0.99, 10000, True, 0, seeds=seeds[:2])
    Hr, _ = pd.ransacH2el_batch(U[:2], 4.0, 0.99, 10000, True, 0, seeds=seeds[:2], raw=True)
    for p in range(2):
        assert np.abs(Hr[p]).sum() > 0 and np.allclose(Hc[p], np.linalg.inv(Hr[p].T), rtol=1e-12, atol=0)
        x1 = np.array([U[p][0, 0], U[p][0, 1], 1.0]); x2 = Hc[p] @ x1     # maps a point of image 1 into image 2 (finite)
        assert np.isfinite(x2).all()


def test_ellipse_ransac_rejects_bad_arguments():
    u, _ = syn.ellipse_pairs(50, 0.5, 1.0, 1, 0.05)
    with pytest.raises(ValueError):
        pd.ransacH2el(u[:, :6])
    with pytest.raises(ValueError):
        pd.ransacH2el(u[:1])
    with pytest.raises(Exception):
        pd.ransacH2el(u, inl_limit=2)
    for bad in (dict(th=0.0), dict(th=-1.0), dict(conf=0.0), dict(conf=1.0), dict(max_iters=0)):      # EINVAL from the library, never a device fault
        with pytest.raises(ValueError):                    # MI_DEGENSAC_EINVAL -> ValueError, as std::invalid_argument in the reference binding
            pd.ransacH2el(u, seed=1, **bad)
