"""GPU: MI_DEGENSAC_FLAG_LEGACY_F — the reference's older fundamental-matrix drivers exp_ransacF / exp_ransacFcustom
(exp_ranF.c:242, :811) through the C-ABI: against the golden fixtures made from the unmodified reference,
against the CPU restatement on seeded cases (single calls and a batch, every kernel variant), and the argument checks."""
import numpy as np
import pytest

import pydegensac_amd as pd
from pydegensac_amd import _lib, synthetic as syn
from tests import golden_util as gu

pytestmark = pytest.mark.gpu
L_FIX = gu.fixtures("L")
ET = {0: "sampson", 1: "symm_epipolar"}


@pytest.mark.parametrize("path", L_FIX, ids=[p.split("/")[-1][:-4] for p in L_FIX])
def test_legacy_drivers_match_reference_goldens(path):
    g = gu.load(path); kw = g["call"]
    F, m = pd.ransacF_legacy(g["p1"], g["p2"], kw.get("px_th", 0.5), kw.get("conf", 0.9999), kw.get("max_iters", 100000),
                             ET[kw.get("error_type", 0)], seed=g["seed"])
    st = pd.last_stats()
    assert (st["samples"], st["lo_runs"], st["I"]) == (g["samples"], g["lo_runs"], g["I"])
    assert np.array_equal(np.asarray(m, bool), g["mask"]) and gu.rel(F, g["model"]) <= 1e-6


def test_legacy_batch_matches_oracle_on_every_variant(oracle_port):
    A, B, seeds = [], [], []
    for i in range(10):
        p1, p2, _, _ = syn.two_view_fundamental([300, 800, 2000][i % 3], 0.4, 0.1, seed=40 + i, plane_fraction=[0.0, 0.6, 0.9][i % 3])
        A.append(p1); B.append(p2); seeds.append(11 + i)
    want = [oracle_port.find_fundamental(a, b, 0.5, 0.9999, 20000, 0, False, 0.0, True, seed=s, legacy=True) for a, b, s in zip(A, B, seeds)]
    usual = [oracle_port.find_fundamental(a, b, 0.5, 0.9999, 20000, 0, False, 0.0, True, seed=s) for a, b, s in zip(A, B, seeds)]
    assert any(w[2]["samples"] != u[2]["samples"] for w, u in zip(want, usual))          # the rule matters on this batch
    for variant in (_lib.TUNE_LATENCY, _lib.TUNE_THROUGHPUT, _lib.TUNE_THROUGHPUT4):
        F, m = pd.ransacF_legacy_batch(A, B, 0.5, 0.9999, 20000, "sampson", seeds=seeds, tuning=variant)
        st = pd.last_stats()
        for p in range(len(A)):
            Fo, mo, so = want[p]
            assert (st[p]["samples"], st[p]["lo_runs"], st[p]["I"]) == (so["samples"], so["lo_runs"], so["I"]), (variant, p)
            assert np.array_equal(np.asarray(m[p], bool), mo) and gu.rel(F[p], Fo) <= 1e-6, (variant, p)


LS_FIX = gu.fixtures("LS")


@pytest.mark.parametrize("path", LS_FIX, ids=[p.split("/")[-1][:-4] for p in LS_FIX])
def test_legacy_symmetric_check_matches_reference_goldens(path):
    """exp_ransacFcustom WITH its symmetric check (all points, 16 th, the final mask filtered with the driver's last model,
    exp_ranF.c:943-953, :1196-1203) against fixtures from the unmodified reference"""
    g = gu.load(path); kw = g["call"]
    for variant in (_lib.TUNE_LATENCY, _lib.TUNE_THROUGHPUT, _lib.TUNE_THROUGHPUT4):
        F, m = pd.ransacF_legacy(g["p1"], g["p2"], kw.get("px_th", 0.5), kw.get("conf", 0.9999), kw.get("max_iters", 100000),
                                 ET[kw.get("error_type", 0)], seed=g["seed"], symmetric_error_check=True, tuning=variant)
        st = pd.last_stats()
        assert (st["samples"], st["lo_runs"], st["I"]) == (g["samples"], g["lo_runs"], g["I"]), variant
        assert np.array_equal(np.asarray(m, bool), g["mask"]) and gu.rel(F, g["model"]) <= 1e-6, variant


def test_legacy_symmetric_check_matches_oracle_on_seeded_cases(oracle_port):
    """... and against the CPU restatement on 36 seeded cases: runs that end in the main loop, right after a local
    optimisation, after the DEGENSAC branch (plane-dominated scenes), few and many samples, both metrics"""
    bad = []
    for i in range(36):
        n = [120, 400, 1000, 2000][i % 4]; pf = [0.0, 0.6, 0.9][i % 3]; et = i % 2; mi = [300, 3000, 20000][(i // 2) % 3]
        p1, p2, _, _ = syn.two_view_fundamental(n, [0.3, 0.5, 0.7][(i // 3) % 3], [0.1, 0.5][(i // 5) % 2], seed=500 + i, plane_fraction=pf)
        Fo, mo, so = oracle_port.find_fundamental(p1, p2, 0.5, 0.9999, mi, et, True, 0.0, True, seed=77 + i, legacy=True)
        F, m = pd.ransacF_legacy(p1, p2, 0.5, 0.9999, mi, ET[et], seed=77 + i, symmetric_error_check=True,
                                 tuning=[_lib.TUNE_LATENCY, _lib.TUNE_THROUGHPUT, _lib.TUNE_THROUGHPUT4][i % 3])
        st = pd.last_stats()
        ok = (st["samples"], st["lo_runs"], st["I"]) == (so["samples"], so["lo_runs"], so["I"]) and np.array_equal(np.asarray(m, bool), mo) and gu.rel(F, Fo) <= 1e-6
        if not ok:
            bad.append((i, n, pf, et, mi, st["samples"], so["samples"], int((np.asarray(m, bool) != mo).sum())))
    assert not bad, bad


def test_legacy_flag_argument_checks():
    p1, p2, _, _ = syn.two_view_fundamental(200, 0.5, 0.1, seed=1)
    l1 = np.concatenate([p1, np.tile([5.0, 0, 0, 5.0], (200, 1))], 1); l2 = np.concatenate([p2, np.tile([5.0, 0, 0, 5.0], (200, 1))], 1)
    with pytest.raises(ValueError):                                  # the legacy drivers take no LAF arguments
        pd.findFundamentalMatrix_(l1, l2, 0.5, 0.9999, 1000, 0, True, 3.0, True, seed=1, flags=_lib.FLAG_LEGACY_F)
    h1, h2, _, _ = syn.homography_pairs(200, 0.5, 0.5, seed=1)
    from pydegensac_amd import api
    with pytest.raises(ValueError):
        api._call_single("H", h1, h2, 1.0, 0.999, 1000, 0, False, 0.0, True, 1, 0, _lib.FLAG_LEGACY_F, 0)
