"""GPU: every kernel variant (512 / 256 / 128 threads per workgroup) and placement mode (0: points + pool in the HBM
workspace, 1: both in LDS, 2: pool in LDS) returns the same bits, and the oracle agrees with them."""
import numpy as np
import pytest

import pydegensac_amd as pd
from pydegensac_amd import _lib, synthetic as syn

pytestmark = pytest.mark.gpu

VARIANT = {512: _lib.TUNE_LATENCY, 256: _lib.TUNE_THROUGHPUT, 128: _lib.TUNE_THROUGHPUT4}
PLACE = {0: _lib.TUNE_PLACE_HBM, 1: _lib.TUNE_PLACE_LDS, 2: _lib.TUNE_PLACE_POOL_LDS}


def tune(variant, mode, seq_pool=False):
    """params.tuning word (include/mi_degensac.h): kernel variant, placement, sampler stage"""
    return VARIANT[variant] | PLACE[mode] | (_lib.TUNE_SEQ_POOL if seq_pool else 0)


def _f_batch():
    A, B = [], []
    for i, (n, pf) in enumerate([(2000, 0.0), (900, 0.0), (1500, 0.7), (300, 0.0)]):
        p1, p2, _, _ = syn.two_view_fundamental(n, 0.4, 0.1, seed=20 + i, plane_fraction=pf); A.append(p1); B.append(p2)
    return A, B


def test_fundamental_variants_and_modes_agree(oracle_port):
    A, B = _f_batch(); seeds = [1, 2, 3, 4]
    ref = None
    for variant in (512, 256, 128):
        for mode in (1, 2, 0):
            F, m = pd.findFundamentalMatrixBatch(A, B, max_iters=20000, seeds=seeds, tuning=tune(variant, mode))
            st = pd.last_stats()
            assert all(s_["threads"] == variant and s_["placement"] == mode for s_ in st), (variant, mode, st[0])
            if ref is None:
                ref = (np.asarray(F).copy(), [np.asarray(x).copy() for x in m])
            else:
                assert np.array_equal(np.asarray(F), ref[0]), (variant, mode)
                assert all(np.array_equal(np.asarray(x), y) for x, y in zip(m, ref[1])), (variant, mode)
    for p in range(len(A)):
        Fo, mo, _ = oracle_port.find_fundamental(A[p], B[p], 0.5, 0.9999, 20000, seed=seeds[p])
        assert np.array_equal(ref[1][p], mo.astype(bool))
        a = ref[0][p].ravel(); b = np.asarray(Fo).ravel()
        assert np.linalg.norm(a - b) <= 1e-9 * np.linalg.norm(b)


def _check_against_oracle(oracle_port, A, B, seeds, F, m, st, tag):
    """every pair of a batch result against the CPU oracle: counters, mask, model"""
    for p in range(len(A)):
        key = (tuple(A[p].shape), int(seeds[p]))
        if key not in _ORACLE_CACHE:
            _ORACLE_CACHE[key] = oracle_port.find_fundamental(A[p], B[p], 0.5, 0.9999, 20000, seed=int(seeds[p]))
        Fo, mo, so = _ORACLE_CACHE[key]
        assert (st[p]["samples"], st[p]["lo_runs"], st[p]["degen"]) == (so["samples"], so["lo_runs"], so["degen"]), (tag, p)
        assert np.array_equal(np.asarray(m[p]), mo.astype(bool)), (tag, p)
        a = np.asarray(F[p]).ravel(); b = np.asarray(Fo).ravel()
        assert np.linalg.norm(a - b) <= 1e-9 * np.linalg.norm(b), (tag, p)


_ORACLE_CACHE = {}


def test_cooperative_helpers_match_the_oracle(oracle_port):
    """cooperative large-n mode forced on a small batch (placement HBM, 1 / 3 / 7 helper workgroups per pair, several pairs
    per owner): every run, pair by pair, against the CPU oracle (counters, masks, models), not against another GPU run.
    The local optimisation's repetitions run as units of their own (stage 4: one claiming workgroup per repetition, speculated
    generator states, in-order replay); with TUNE_F_SERIAL_REPS they run one after the other with every pass distributed."""
    A, B = _f_batch(); A = A * 3; B = B * 3; seeds = list(range(1, 13))
    for variant in (512, 256, 128):
        for helpers, dist, serial in ((255, 0, 0), (1, 0, 0), (3, 1, 0), (7, 0, 0), (7, 1, 0), (3, 1, 1), (7, 0, 1), (12, 0, 0)):    # dist: the LO's full passes distributed too (TUNE_COOP_ALL_PASSES)
            F, m = pd.findFundamentalMatrixBatch(A, B, max_iters=20000, seeds=seeds, tuning=tune(variant, 0) | _lib.TUNE_HELPERS(helpers) | (dist * _lib.TUNE_COOP_ALL_PASSES)
                                                 | (serial * _lib.TUNE_F_SERIAL_REPS))
            _check_against_oracle(oracle_port, A, B, seeds, F, m, pd.last_stats(), (variant, helpers, dist, serial))


def test_repetitions_one_per_wave_equal_the_serial_order(oracle_port):
    """The local optimisation (exp_inFranicustom, exp_ranF.c:745-806) and the DEGENSAC branch's innerH (ranH.c:18-135) run
    their ten repetitions one per wave from speculated generator states and commit them in order (dg_inFrani_waves,
    dg_innerH_waves); the serial order is kept behind TUNE_F_SERIAL_REPS.  Plane-dominated scenes (DEGENSAC events, Ih > 0)
    and ordinary ones (several LO runs), every workgroup size = rounds of 8 / 4 / 2 repetitions, placements LDS and
    workspace: both orders against the CPU oracle pair by pair (counters incl. every kind of pass, masks, models)."""
    A, B = [], []
    for i, (n, pf, ir) in enumerate([(1500, 0.7, 0.4), (400, 0.9, 0.5), (2000, 0.6, 0.4), (800, 0.8, 0.3), (1200, 0.5, 0.6), (300, 0.95, 0.6),
                                     (2000, 0.0, 0.4), (900, 0.0, 0.25), (150, 0.0, 0.5), (40, 0.0, 0.8)]):
        p1, p2, _, _ = syn.two_view_fundamental(n, ir, 0.1, seed=70 + i, plane_fraction=pf); A.append(p1); B.append(p2)
    seeds = [11, 12, 13, 14, 15, 16, 17, 18, 19, 20]
    ora = [oracle_port.find_fundamental(A[p], B[p], 0.5, 0.9999, 20000, seed=seeds[p]) for p in range(len(A))]
    assert sum(o[2]["Ih"] > 0 for o in ora) >= 4 and sum(o[2]["lo_runs"] >= 2 for o in ora) >= 4, "the scenes must reach innerH and run several local optimisations"
    for variant in (512, 256, 128):
        for mode in (1, 0):
            for serial in (0, _lib.TUNE_F_SERIAL_REPS):
                F, m = pd.findFundamentalMatrixBatch(A, B, max_iters=20000, seeds=seeds, tuning=tune(variant, mode) | serial)
                st = pd.last_stats()
                for p in range(len(A)):
                    Fo, mo, so = ora[p]
                    key = lambda s_: (s_["samples"], s_["lo_runs"], s_["degen"], s_["Ih"], s_["models"])
                    assert key(st[p]) == key(so), (variant, mode, serial, p, st[p], so)
                    assert (st[p]["h_passes"], st[p]["full_passes"], st[p]["ex_passes"]) == (so["hds_passes"], so["full_passes"], so["ex_passes"]), (variant, mode, serial, p)
                    assert np.array_equal(np.asarray(m[p]), mo.astype(bool)), (variant, mode, serial, p)
                    a = np.asarray(F[p]).ravel(); b = np.asarray(Fo).ravel()
                    assert np.linalg.norm(a - b) <= 1e-9 * np.linalg.norm(b), (variant, mode, serial, p)


def test_setting_pairs_aside_matches_the_oracle(oracle_port):
    """Pairs are written back to their workspace after the discovery round and resumed by priority (dg_args::park_sam,
    park_long).  Forced on a small batch: 4 resident workgroups for 24 pairs, threshold = one chunk / four chunks / 2048
    samples, two "many samples left" factors, all three workgroup sizes and every placement: every run, pair by pair,
    against the CPU oracle; the set-aside pairs are counted (the factors put nearly all of them into the "many samples
    left" queue in one setting and all of them into the other queue in another)."""
    A, B = _f_batch(); A = A * 6; B = B * 6; seeds = list(range(1, 25))
    for variant in (256, 512, 128):
        for mode in (2, 1, 0):
            for park, lg in ((255, 0), (1, 0), (4, 1), (8, 3), (8, 7)):          # 255 = off; else units of 256 samples; lg = TUNE_LONG_SHIFT
                F, m = pd.findFundamentalMatrixBatch(A, B, max_iters=20000, seeds=seeds,
                                                     tuning=tune(variant, mode) | _lib.TUNE_LONG_SHIFT(lg) | _lib.TUNE_HELPERS(255) | _lib.TUNE_SET_ASIDE(park) | _lib.TUNE_GRID_CAP(4))
                st = pd.last_stats()
                aside = sum(s_["set_aside"] for s_ in st)
                assert (aside == 0) if park == 255 else (aside >= 4), (variant, mode, park, aside)
                _check_against_oracle(oracle_port, A, B, seeds, F, m, st, (variant, mode, park, lg))


def test_homography_variants_and_modes_agree(oracle_port):
    """every workgroup size and placement returns the same bits, and those are the CPU oracle's (masks and counters exact,
    raw model to 1e-9)"""
    A, B = [], []
    for i, n in enumerate([1200, 400, 2500]):
        p1, p2, _, _ = syn.homography_pairs(n, 0.4, 0.5, seed=30 + i, laf=True); A.append(p1); B.append(p2)
    ora = [oracle_port.find_homography(A[p], B[p], 1.0, 0.999, 20000, 0, True, 3.0, seed=5 + p) for p in range(3)]
    ref = None
    for variant in (512, 256, 128):
        for mode in (1, 2, 0):
            H, m = pd.findHomographyBatch(A, B, 1.0, 0.999, 20000, 3.0, "sampson", True, seeds=[5, 6, 7], tuning=tune(variant, mode))
            st = pd.last_stats()
            assert all(s_["threads"] == variant for s_ in st)      # n = 2500 does not fit "both in LDS" at 512 threads
            for p in range(3):
                Ho, mo, so = ora[p]
                assert (st[p]["samples"], st[p]["lo_runs"], st[p]["rejected"], st[p]["models"]) == (so["samples"], so["lo_runs"], so["rejected"], so["models"]), (variant, mode, p)
                assert np.array_equal(np.asarray(m[p]), mo), (variant, mode, p)
                if np.abs(Ho).sum() > 0:      # the batch API returns inv(raw.T) (utils.py:108)
                    Hu = np.linalg.inv(np.asarray(Ho).T)
                    assert np.linalg.norm(np.asarray(H[p]) - Hu) <= 1e-8 * np.linalg.norm(Hu), (variant, mode, p)
            if ref is None:
                ref = (np.asarray(H).copy(), [np.asarray(x).copy() for x in m])
            else:
                assert np.array_equal(np.asarray(H), ref[0]), (variant, mode)
                assert all(np.array_equal(np.asarray(x), y) for x, y in zip(m, ref[1])), (variant, mode)


def test_homography_lo_one_repetition_per_wave_equals_the_serial_order(oracle_port):
    """The homography kernel runs the ten repetitions of a local optimisation on one wave each and replays the hash
    table / best-so-far / errs[] rotation in repetition order.  Forcing the reference's serial order
    (TUNE_H_SERIAL_LO) must give the same bits and counters for every workgroup size (2, 4, 8 waves = 5, 3, 2 rounds), with and
    without the symmetric metrics, and the oracle must agree."""
    A, B = [], []
    for i, n in enumerate([5000, 700, 2500, 64, 20, 9]):
        p1, p2, _, _ = syn.homography_pairs(n, 0.45, 0.5, seed=130 + i, laf=True); A.append(p1); B.append(p2)
    seeds = [11, 12, 13, 14, 15, 16]
    for err in ("sampson", "symm_max"):
        ref = None
        for variant in (512, 256, 128):
            for serial in (0, _lib.TUNE_H_SERIAL_LO):
                H, m = pd.findHomographyBatch(A, B, 1.5, 0.999, 20000, 3.0, err, True, seeds=seeds, tuning=VARIANT[variant] | serial)
                st = pd.last_stats()
                cur = (np.asarray(H).copy(), [np.asarray(x).copy() for x in m], [(s_["samples"], s_["lo_runs"], s_["models"], s_["I"]) for s_ in st])
                if ref is None:
                    ref = cur
                else:
                    assert np.array_equal(cur[0], ref[0]), (err, variant, serial)
                    assert all(np.array_equal(x, y) for x, y in zip(cur[1], ref[1])), (err, variant, serial)
                    assert cur[2] == ref[2], (err, variant, serial)
        assert any(c[1] >= 2 for c in ref[2])                     # several local optimisations share one hash table
        et = {"sampson": 0, "symm_max": 2}[err]
        for p in (0, 1, 3, 5):
            Ho, mo, so = oracle_port.find_homography(A[p], B[p], 1.5, 0.999, 20000, et, True, 3.0, seed=seeds[p])
            assert (ref[2][p][0], ref[2][p][1]) == (so["samples"], so["lo_runs"]), (err, p)
            assert np.array_equal(ref[1][p], mo.astype(bool)), (err, p)
            if np.abs(np.asarray(Ho)).sum() == 0:
                assert np.abs(ref[0][p]).sum() == 0, (err, p)
                continue
            Hu = np.linalg.inv(np.asarray(Ho).reshape(3, 3).T)         # utils.py:108
            assert np.linalg.norm(ref[0][p] - Hu) <= 1e-6 * np.linalg.norm(Hu), (err, p)


def test_sequential_pool_stage_fallback_matches_goldens():
    """The parallel pool-swap stage relies on the LDS exchange order that the library probes once per device
    (mi_degensac_pool_stage_parallel); when the probe fails every launch uses the sequential stage.  Forcing that
    fallback must reproduce the reference goldens, and the device this suite runs on must pass the probe."""
    import ctypes as C
    from tests import golden_util as gu
    assert _lib.lib().mi_degensac_pool_stage_parallel(0) == 1
    for path in gu.fixtures("F")[:6] + gu.fixtures("H")[:6]:
        g = gu.load(path); kw = g["call"]
        if g["n"] <= 10:
            continue
        for variant in (512, 256, 128):
            t = VARIANT[variant] | _lib.TUNE_SEQ_POOL
            if g["kind"] == "F":
                M, m = pd.findFundamentalMatrix_(g["p1"], g["p2"], kw.get("px_th", 0.5), kw.get("conf", 0.9999), kw.get("max_iters", 100000),
                                                 kw.get("error_type", 0), kw.get("sym_check", True), kw.get("laf_coef", 0.0),
                                                 kw.get("degen", True), seed=g["seed"], tuning=t)
            else:
                M, m = pd.findHomography_(g["p1"], g["p2"], kw.get("px_th", 1.0), kw.get("conf", 0.999), kw.get("max_iters", 50000),
                                          kw.get("error_type", 0), kw.get("sym_check", True), kw.get("laf_coef", 0.0), seed=g["seed"], tuning=t)
            st = pd.last_stats()
            assert st["samples"] == g["samples"] and st["lo_runs"] == g["lo_runs"], path
            if np.abs(g["model"]).sum():
                assert np.array_equal(np.asarray(m), g["mask"]) and gu.rel(M, g["model"]) < 1e-6, path
    # the unit-level sample stream, both stages
    out = np.zeros((600, 7), np.int32); out2 = np.zeros((600, 7), np.int32)
    L = _lib.lib()
    _lib.check(L.mi_degensac_sample_stream_ex(777, 2000, 7, 600, 0, 0, out.ctypes.data_as(C.POINTER(C.c_int32))))
    _lib.check(L.mi_degensac_sample_stream_ex(777, 2000, 7, 600, 0, 1, out2.ctypes.data_as(C.POINTER(C.c_int32))))
    assert np.array_equal(out, out2)


def test_cooperative_owner_runs_several_pairs_in_a_row(oracle_port):
    """more pairs than owner slots (512 threads: 256 resident workgroups / 24 = 10 owners for 30 pairs): every owner and its helpers
    go through several pairs one after the other — the per-pair state of the cooperative mode (model table in use, early solves,
    the deep sampler pipeline's pending draws, the draws assumed by the local optimisation's stage 4) must start afresh; pair by pair
    against the CPU oracle"""
    A, B = [], []
    for i in range(30):
        p1, p2, _, _ = syn.two_view_fundamental(600 + 37 * (i % 40), 0.35 + 0.01 * (i % 20), 0.1, seed=200 + i, plane_fraction=0.6 if i % 5 == 0 else 0.0)
        A.append(p1); B.append(p2)
    seeds = list(range(11, 41))
    for variant, helpers in ((512, 23), (256, 23), (128, 7)):
        F, m = pd.findFundamentalMatrixBatch(A, B, max_iters=20000, seeds=seeds, tuning=tune(variant, 0) | _lib.TUNE_HELPERS(helpers))
        _check_against_oracle(oracle_port, A, B, seeds, F, m, pd.last_stats(), (variant, helpers, "several pairs per owner"))
