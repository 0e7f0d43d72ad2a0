"""CPU suite (-m "not gpu"): pins the oracle.

  * the restatement (oracle/dg_oracle.c) against the golden fixtures generated from the unmodified
    reference (tests/golden, made by tests/golden/make_golden.py) — trajectory counters and mask exact,
    model <= 1e-9 relative;
  * the restatement against oracle/_ref live, when that build is present (this container);
  * unit pieces against libc / the reference's own functions.
"""
import ctypes as C
import os

import numpy as np
import pytest

from pydegensac_amd import synthetic as syn
from tests import golden_util as gu


@pytest.mark.parametrize("path", gu.fixtures("F"), ids=lambda p: os.path.basename(p)[:-4])
def test_port_matches_golden_fundamental(oracle_port, path):
    g = gu.load(path)
    F, m, st = oracle_port.find_fundamental(g["p1"], g["p2"], seed=g["seed"], **g["call"])
    assert st["samples"] == g["samples"] and st["lo_runs"] == g["lo_runs"]
    assert st["full_passes"] == g["full_passes"] and st["ex_passes"] == g["ex_passes"]
    assert np.array_equal(m, g["mask"]) or np.abs(g["model"]).sum() == 0
    assert gu.rel(F, g["model"]) < 1e-9


@pytest.mark.parametrize("path", gu.fixtures("H"), ids=lambda p: os.path.basename(p)[:-4])
def test_port_matches_golden_homography(oracle_port, path):
    g = gu.load(path)
    H, m, st = oracle_port.find_homography(g["p1"], g["p2"], seed=g["seed"], **g["call"])
    if g["n"] <= 10:
        pytest.skip("4-point u2h path of the reference reads uninitialised memory (Htools.c:108-114)")
    assert (st["samples"], st["lo_runs"], st["rejected"], st["models"]) == (g["samples"], g["lo_runs"], g["rejected"], g["full_passes"])
    if np.abs(g["model"]).sum() != 0:
        assert np.array_equal(m, g["mask"])
        assert gu.rel(H, g["model"]) < 1e-8


def test_rng_matches_libc(oracle_port):
    libc = C.CDLL("libc.so.6")
    libc.srand.argtypes = [C.c_uint]; libc.rand.restype = C.c_int
    for seed in [0, 1, 2, 12345, 2**31 - 1, 2**31, 2**32 - 1, 987654321]:
        libc.srand(seed)
        want = [libc.rand() for _ in range(50)]
        got = np.zeros(50, np.int32)
        oracle_port.lib().dg_oracle_rand_stream(C.c_uint(seed), 50, oracle_port.ip(got))
        assert list(got) == want, seed


def test_units_match_reference(oracle_port, oracle_ref):
    R = oracle_ref.lib(); P = oracle_port.lib(); dp = oracle_port.dp; ip = oracle_port.ip
    rng = np.random.default_rng(0)
    p1, p2, lab, _ = syn.two_view_fundamental(800, 0.5, 0.1, seed=1)
    n = 800
    u = np.ones((n, 6)); u[:, 0:2] = p1; u[:, 3:5] = p2
    F = rng.normal(size=9)
    for name_r, name_p in [("FDs", "dg_oracle_FDs"), ("FDsSym", "dg_oracle_FDsSym")]:
        d1 = np.zeros(n); d2 = np.zeros(n)
        getattr(R, name_r)(dp(u), dp(F), dp(d1), n); getattr(P, name_p)(dp(u), dp(F), dp(d2), n)
        assert np.array_equal(d1, d2)
    # nullspace / slcm / rroots3 on real samples
    for t in range(50):
        ids = rng.choice(n, 7, replace=False)
        A = np.zeros(81)
        for i, q in enumerate(ids):
            A[9 * i:9 * i + 9] = np.outer(u[q, 3:6], u[q, 0:3]).ravel()
        A1 = A.copy(); A2 = A.copy(); s1 = np.zeros(81); s2 = np.zeros(81); buf = np.zeros(18, np.int32)
        r1 = R.nullspace(dp(A1), dp(s1), 9, ip(buf)); r2 = P.dg_oracle_nullspace(dp(A2), dp(s2), 9)
        assert r1 == r2 == 2 and np.array_equal(s1[:18], s2[:18])
        pa = np.zeros(4); pb = np.zeros(4); b1 = s1[9:18].copy(); b2 = s2[9:18].copy()
        R.slcm(dp(s1), dp(b1), dp(pa)); P.dg_oracle_slcm(dp(s2), dp(b2), dp(pb))
        assert np.array_equal(pa, pb) and np.array_equal(b1, b2)
        ra = np.zeros(3); rb = np.zeros(3)
        assert R.rroots3(dp(pa), dp(ra)) == P.dg_oracle_rroots3(dp(pb), dp(rb)) and np.array_equal(ra, rb)
    # svduv bit-exact, eig / u2f / u2h to rounding (external LAPACK in the reference)
    for shape in [(9, 8), (3, 3)]:
        A = rng.normal(size=shape)
        a1 = A.copy().ravel(); a2 = A.copy().ravel()
        d1 = np.zeros(9); d2 = np.zeros(9); u1 = np.zeros(81); u2 = np.zeros(81); v1 = np.zeros(64); v2 = np.zeros(64)
        R.svduv(dp(d1), dp(a1), dp(u1), shape[0], dp(v1), shape[1]); P.dg_oracle_svduv(dp(d2), dp(a2), dp(u2), shape[0], dp(v2), shape[1])
        assert np.array_equal(u1, u2) and np.array_equal(v1, v2) and np.array_equal(d1, d2)
    inl = np.nonzero(lab)[0].astype(np.int32)
    buf = np.zeros(18 * n)
    for ln in [8, 10, 14, 300]:
        sel = np.ascontiguousarray(rng.choice(inl, ln, replace=False).astype(np.int32))
        F1 = np.zeros(9); F2 = np.zeros(9)
        R.u2f(dp(u), ip(sel), ln, dp(F1), dp(buf)); P.dg_oracle_u2f(dp(u), ip(sel), ln, dp(F2))
        assert np.linalg.norm(F1 - F2) < 1e-8 * np.linalg.norm(F1)          # same sign, same scale
    for ln in [5, 10, 12, 200]:
        sel = np.ascontiguousarray(rng.choice(inl, ln, replace=False).astype(np.int32))
        H1 = np.zeros(9); H2 = np.zeros(9)
        R.u2h(dp(u), ip(sel), ln, dp(H1), dp(buf)); P.dg_oracle_u2h(dp(u), ip(sel), ln, dp(H2))
        assert np.linalg.norm(H1 - H2) < 1e-7 * np.linalg.norm(H1)
    # hash
    lst = np.ascontiguousarray(rng.choice(5000, 321, replace=False).astype(np.int32))
    R.SuperFastHash.restype = C.c_uint32
    assert R.SuperFastHash(lst.ctypes.data_as(C.c_char_p), 321 * 4) == P.dg_oracle_hash(ip(lst), 321)
    for a in [(801, 2000, 7, 0.9999), (10, 2000, 7, 0.99), (2000, 2000, 4, 0.999), (3, 100, 7, 0.5)]:
        assert R.nsamples(*[C.c_int(x) for x in a[:3]], C.c_double(a[3])) == P.dg_oracle_nsamples(*[C.c_int(x) for x in a[:3]], C.c_double(a[3]))


@pytest.mark.parametrize("dseed,plane", [(0, 0.0), (1, 0.7), (2, 0.0)])
def test_port_matches_reference_live(oracle_port, oracle_ref, dseed, plane):
    p1, p2, _, _ = syn.two_view_fundamental(1200, 0.4, 0.1, seed=dseed, plane_fraction=plane)
    mi = 100000 if plane == 0 else 2000
    for seed in [3, 11]:
        F, m, st = oracle_ref.find_fundamental(p1, p2, seed=seed, count_models=True, max_iters=mi)
        F2, m2, st2 = oracle_port.find_fundamental(p1, p2, seed=seed, max_iters=mi)
        assert (st["samples"], st["lo_runs"], st["full_passes"], st["ex_passes"]) == \
               (st2["samples"], st2["lo_runs"], st2["full_passes"], st2["ex_passes"])
        assert np.array_equal(m, m2) and gu.rel(F, F2) < 1e-9


def test_port_matches_reference_random_sweep(oracle_port, oracle_ref):
    """Randomised F / H problems (every metric, LAF on/off, plane-dominated scenes, n = 8..3000): the restatement and the
    unmodified reference build agree on masks and sample / LO counters (tools/cpu_port_vs_ref.py runs longer sweeps and
    documents the three known exceptions through the reference's uninitialised-memory paths; this seed has none)."""
    import os, sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import cpu_port_vs_ref
    bad, loose, worst, _ = cpu_port_vs_ref.run(150, 5, verbose=True)
    assert bad == 0
    assert worst < 1e-5          # ill-conditioned (plane-dominated) final fits may differ beyond rounding, never grossly


@pytest.mark.parametrize("path", gu.fixtures("E"), ids=lambda p: os.path.basename(p)[:-4])
def test_port_matches_golden_ellipse_ransac(oracle_port, path):
    """ransacH2el (ranH2el.c:19): the restatement against fixtures from the unmodified reference"""
    g = gu.load(path)
    H, m, st = oracle_port.ransacH2el(g["p1"], seed=g["seed"], **g["call"])
    assert (st["samples"], st["lo_runs"], st["I"]) == (g["samples"], g["lo_runs"], g["I"])
    assert np.array_equal(m, g["mask"])
    assert gu.rel(H, g["model"]) < 1e-9


def test_ellipse_ransac_port_matches_reference_live(oracle_port, oracle_ref):
    """ransacH2el restated against the reference build on seeded problems: LO on / off, a finite inlLimit (random subsets
    inside iterH), 60..3000 correspondences, 10-50 % inliers"""
    for n, ir, sig, ln in [(1000, 0.3, 1.0, 0.05), (3000, 0.1, 1.0, 0.05), (400, 0.15, 1.5, 0.1), (60, 0.5, 1.0, 0.05)]:
        for seed in (2, 5):
            u, _ = syn.ellipse_pairs(n, ir, sig, seed + n, ln)
            for lo, lim, mi, th in [(True, 0, 10000, 4.0), (False, 0, 3000, 4.0), (True, 25, 10000, 9.0)]:
                Hr, mr, sr = oracle_ref.ransacH2el(u, th, 0.99, mi, lo, lim, seed)
                Hp, mp, sp = oracle_port.ransacH2el(u, th, 0.99, mi, lo, lim, seed)
                assert (sr["samples"], sr["lo_runs"], sr["I"]) == (sp["samples"], sp["lo_runs"], sp["I"]), (n, seed, lo, lim)
                assert np.array_equal(mr, mp) and gu.rel(Hr, Hp) < 1e-9, (n, seed, lo, lim)


def test_h_symmetric_metrics_match_reference(oracle_port, oracle_ref):
    """the four symmetric transfer errors of the restatement (HDS_full kinds 1..4) against the reference's own
    HDsSymMaxSq / HDsSymMax / HDsSymSumSq / HDsSymSum (Htools.c:202-370), bit for bit"""
    R = oracle_ref.lib(); P = oracle_port.lib(); dp = oracle_port.dp
    p1, p2, lab, Hgt = syn.homography_pairs(n=700, inlier_ratio=0.5, sigma=0.5, seed=3)
    n = 700
    u = np.ones((n, 6)); u[:, 0:2] = p1; u[:, 3:5] = p2
    rng = np.random.default_rng(5)
    Hc = np.linalg.inv(Hgt).T.ravel().copy()          # the driver's column-wise image2->image1 form (utils.py:108)
    for H in [Hc, Hc * (1 + 0.01 * rng.normal(size=9)), rng.normal(size=9)]:
        H = np.ascontiguousarray(H)
        for kind, name in [(1, "HDsSymMaxSq"), (2, "HDsSymMax"), (3, "HDsSymSumSq"), (4, "HDsSymSum")]:
            d1 = np.zeros(n); d2 = np.zeros(n)
            getattr(R, name)(None, dp(u), dp(H), dp(d1), n)      # lin (Z) is unused by these metrics
            P.dg_oracle_HDS_full(kind, dp(u), dp(H), dp(d2), n)
            assert np.array_equal(d1, d2), name


def test_sampson_bound_of_the_homography_screen_is_a_superset(oracle_port):
    """The homography kernel's main loop skips the exact score of a model whose division-free candidate count (dg_geom.h:
    dg_HDs_maybe_below, restated here operation by operation) does not exceed the bound to beat.  That is only sound if every point
    the reference's HDs (Htools.c:161-200, through pinvJ) puts below the threshold is a candidate: random, scaled, near-singular and
    rank-deficient models, points on a line / at the origin / far away."""
    rng = np.random.default_rng(5)

    def maybe_below(H, u0, u1, u3, u4, tb):
        w = H[2] * u3 + H[5] * u4 + H[8]
        r1 = (H[0] * u3 + H[3] * u4 + H[6]) - u0 * w; r2 = (H[1] * u3 + H[4] * u4 + H[7]) - u1 * w
        a = H[0] - H[2] * u0; b = H[3] - H[5] * u0; d = H[1] - H[2] * u1; e = H[4] - H[5] * u1
        cc = w * w
        m11 = a * a + b * b + cc; m22 = d * d + e * e + cc; m12 = a * d + b * e
        det = m11 * m22 - m12 * m12; q = m22 * r1 * r1 - 2 * m12 * r1 * r2 + m11 * r2 * r2
        with np.errstate(invalid="ignore"):
            well = det > 1e-7 * (m11 * m22)
            return ~(well & (q > tb * det))

    n = 4000; checked = 0
    for trial in range(60):
        scale = [1.0, 1e-3, 1e3, 1e-6][trial % 4]
        H = rng.normal(size=9) * scale
        if trial % 5 == 1: H[6:9] = 0                                   # rank-deficient
        if trial % 5 == 2: H = np.eye(3).ravel() + 1e-9 * rng.normal(size=9)   # identity-like: many points are inliers
        if trial % 5 == 3: H[2] = H[5] = 0; H[8] = 1e-12                 # affine with a vanishing last entry
        p1 = rng.uniform(-2000, 2000, size=(n, 2)); p2 = rng.uniform(-2000, 2000, size=(n, 2))
        if trial % 3 == 0: p2 = p1 + rng.normal(size=(n, 2))              # close pairs (inliers of identity-like models)
        if trial % 7 == 0: p1[:, 1] = 2 * p1[:, 0] + 1                    # points on a line
        p1[:5] = 0; p2[:5] = 0
        u = np.ones((n, 6)); u[:, 0:2] = p1; u[:, 3:5] = p2
        d = np.zeros(n)
        oracle_port.lib().dg_oracle_HDs(oracle_port.dp(u), oracle_port.dp(H.copy()), oracle_port.dp(d), n)
        for t in (0.25, 4.0, 400.0):
            t94 = t * 9 / 4
            inl = d < t94                                                  # NaN residuals compare false in the reference as well
            cand = maybe_below(H, p1[:, 0], p1[:, 1], p2[:, 0], p2[:, 1], t94 * (1.0 + 1e-6))
            assert not np.any(inl & ~cand), (trial, t, int(np.sum(inl & ~cand)))
            checked += int(inl.sum())
    assert checked > 10000                                                 # the check saw real inliers, not only empty sets
