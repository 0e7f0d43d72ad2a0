"""CPU: the reference's older fundamental-matrix drivers exp_ransacF (exp_ranF.c:242) and exp_ransacFcustom (:811, without
its symmetric check) — SURVEY 8f #4.  The restatement's legacy rule (oracle/dg_oracle.c: the sample budget follows every
new best model) against the golden fixtures made from the unmodified reference (tests/golden/L_*.npz), and live against
oracle/_ref where it is built."""
import numpy as np
import pytest

from tests import golden_util as gu

L_FIX = gu.fixtures("L")
LS_FIX = gu.fixtures("LS")


@pytest.mark.parametrize("path", L_FIX, ids=[p.split("/")[-1][:-4] for p in L_FIX])
def test_port_legacy_matches_reference_goldens(oracle_port, path):
    g = gu.load(path); kw = g["call"]
    F, m, st = oracle_port.find_fundamental(g["p1"], g["p2"], kw.get("px_th", 0.5), kw.get("conf", 0.9999), kw.get("max_iters", 100000),
                                            kw.get("error_type", 0), False, 0.0, True, seed=g["seed"], legacy=True)
    assert (st["samples"], st["lo_runs"], st["I"]) == (g["samples"], g["lo_runs"], g["I"])
    assert np.array_equal(m, g["mask"]) and gu.rel(F, g["model"]) <= 1e-6


@pytest.mark.parametrize("path", LS_FIX, ids=[p.split("/")[-1][:-4] for p in LS_FIX])
def test_port_legacy_symmetric_check_matches_reference_goldens(oracle_port, path):
    """exp_ransacFcustom with doSymCheck: all points, CHECK_COEF * th, and the final mask filtered on whatever model the
    driver computed last (exp_ranF.c:943-953, :1196-1203) — restated in the oracle only"""
    g = gu.load(path); kw = g["call"]
    F, m, st = oracle_port.find_fundamental(g["p1"], g["p2"], 0.5, 0.9999, 100000, kw.get("error_type", 0), True, 0.0, True, seed=g["seed"], legacy=True)
    assert (st["samples"], st["lo_runs"], st["I"]) == (g["samples"], g["lo_runs"], g["I"])
    assert np.array_equal(m, g["mask"]) and gu.rel(F, g["model"]) <= 1e-6


def test_exp_ransacF_equals_exp_ransacFcustom_and_differs_from_the_LAF_driver(oracle_ref):
    """the two legacy drivers are the same algorithm (Sampson, no symmetric check); the LAF driver keeps its full sample
    budget when the DEGENSAC branch finds the model (exp_ranF.c:1568-1573 is inside the LO block), the legacy ones do not"""
    from pydegensac_amd import synthetic as syn
    differs = 0
    for seed in range(12):
        p1, p2, _, _ = syn.two_view_fundamental(500 + 100 * (seed % 4), 0.4, 0.1, seed=seed, plane_fraction=[0.0, 0.6, 0.9][seed % 3])
        F0, m0, s0 = oracle_ref.find_fundamental_legacy(0, p1, p2, 0.5, 0.9999, 20000, 0, False, seed=seed + 1)
        F1, m1, s1 = oracle_ref.find_fundamental_legacy(1, p1, p2, 0.5, 0.9999, 20000, 0, False, seed=seed + 1)
        assert np.array_equal(m0, m1) and np.array_equal(F0, F1) and s0 == s1
        F2, m2, s2 = oracle_ref.find_fundamental(p1, p2, 0.5, 0.9999, 20000, 0, False, 0.0, True, seed=seed + 1)
        differs += s2["samples"] != s0["samples"]
    assert differs >= 2


def test_port_legacy_matches_reference_live(oracle_port, oracle_ref):
    from pydegensac_amd import synthetic as syn
    for seed in range(24):
        p1, p2, _, _ = syn.two_view_fundamental(300 + 100 * (seed % 6), 0.4, 0.1, seed=100 + seed, plane_fraction=[0.0, 0.0, 0.6, 0.9][seed % 4])
        et = seed % 2
        for sym in (False, True):
            F0, m0, s0 = oracle_ref.find_fundamental_legacy(0, p1, p2, 0.5, 0.9999, 20000, et, sym, seed=seed + 1)
            F1, m1, s1 = oracle_port.find_fundamental(p1, p2, 0.5, 0.9999, 20000, et, sym, 0.0, True, seed=seed + 1, legacy=True)
            assert np.array_equal(m0, m1) and gu.rel(F0, F1) <= 1e-6 and (s0["samples"], s0["lo_runs"]) == (s1["samples"], s1["lo_runs"]), (seed, sym)
