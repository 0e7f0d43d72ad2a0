"""GPU: what round 5 added to the boundary — one batch over several devices from one process, per-call scheduling flags,
hand-over time-outs that discard and re-run instead of failing (or returning invalid results), the homography screen on gfx950."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest

import pydegensac_amd as pd
from pydegensac_amd import _lib, synthetic as syn

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _f_batch(P=11):
    A, B = [], []
    for i in range(P):
        n = [300, 2000, 800, 1500, 64, 1000][i % 6]
        p1, p2, _, _ = syn.two_view_fundamental(n, 0.4, 0.1, seed=900 + i, plane_fraction=0.7 if i % 4 == 1 else 0.0); A.append(p1); B.append(p2)
    return A, B, [77 + 3 * i for i in range(P)]


def _key(st):
    return [(x["samples"], x["lo_runs"], x["models"], x["degen"], x["I"], x["best_sample"]) for x in st]


def test_one_batch_over_a_device_list_equals_one_device(oracle_port):
    """devices=[0, 0] / [0, 0, 0]: contiguous shards as parallel.shard_range, one host thread per entry, host gather.
    (One GPU here, listed several times: two or three concurrent launches on it.)  Bit-identical to the one-shard call."""
    A, B, seeds = _f_batch(11)
    F0, m0 = pd.findFundamentalMatrixBatch(A, B, max_iters=20000, seeds=seeds); s0 = pd.last_stats()
    for devs in ([0], [0, 0], [0, 0, 0], [0] * 12):
        F1, m1 = pd.findFundamentalMatrixBatch(A, B, max_iters=20000, seeds=seeds, devices=devs); s1 = pd.last_stats()
        assert np.array_equal(np.asarray(F0), np.asarray(F1)), devs
        assert all(np.array_equal(np.asarray(a), np.asarray(b)) for a, b in zip(m0, m1)), devs
        assert _key(s0) == _key(s1), devs
    for p in (1, 5, 10):
        Fo, mo, so = oracle_port.find_fundamental(A[p], B[p], 0.5, 0.9999, 20000, seed=seeds[p])
        assert np.array_equal(np.asarray(m0[p]), mo) and s0[p]["samples"] == so["samples"]
    Hs = [syn.homography_pairs(n, 0.4, 0.5, seed=50 + i, laf=True)[:2] for i, n in enumerate([500, 3000, 1200, 800, 2000])]
    HA = [h[0] for h in Hs]; HB = [h[1] for h in Hs]
    H0, k0 = pd.findHomographyBatch(HA, HB, 2.0, 0.999, 20000, 3.0, seeds=[5, 6, 7, 8, 9])
    H1, k1 = pd.findHomographyBatch(HA, HB, 2.0, 0.999, 20000, 3.0, seeds=[5, 6, 7, 8, 9], devices=[0, 0])
    assert np.array_equal(np.asarray(H0), np.asarray(H1)) and all(np.array_equal(np.asarray(a), np.asarray(b)) for a, b in zip(k0, k1))
    with pytest.raises(_lib.MiDegensacError):
        pd.findFundamentalMatrixBatch(A, B, seeds=seeds, devices=[0, 63])           # no such device: ENODEV, nothing half-written is returned
    with pytest.raises(ValueError):
        pd.findFundamentalMatrixBatch(A, B, seeds=seeds, devices=[])


def test_scheduling_flags_are_per_call(oracle_port):
    """MI_DEGENSAC_FLAG_NO_STREAM / _STREAM_ON / _NO_HJOB choose for ONE call, whatever the process-wide default says."""
    A, B, seeds = _f_batch(8)
    prev = _lib.set_stream_mode(0)                          # process default: off
    try:
        F0, m0 = pd.findFundamentalMatrixBatch(A, B, max_iters=30000, seeds=seeds); s0 = pd.last_stats()
        assert sum(s_["streamed"] for s_ in s0) == 0
        F1, m1 = pd.findFundamentalMatrixBatch(A, B, max_iters=30000, seeds=seeds, flags=_lib.FLAG_STREAM_ON | _lib.FLAG_STREAM_TEST(2)); s1 = pd.last_stats()
        assert sum(s_["streamed"] for s_ in s1) >= 1
        _lib.set_stream_mode(1)                             # process default: on; the call says no
        F2, m2 = pd.findFundamentalMatrixBatch(A, B, max_iters=30000, seeds=seeds, flags=_lib.FLAG_NO_STREAM); s2 = pd.last_stats()
        assert sum(s_["streamed"] for s_ in s2) == 0
    finally:
        _lib.set_stream_mode(prev)
    for F, m, s in ((F1, m1, s1), (F2, m2, s2)):
        assert np.array_equal(np.asarray(F0), np.asarray(F)) and all(np.array_equal(np.asarray(a), np.asarray(b)) for a, b in zip(m0, m)) and _key(s0) == _key(s)
    p1, p2, _, _ = syn.homography_pairs(3000, 0.4, 0.5, seed=4, laf=True)
    Ha, ka = pd.findHomographyBatch([p1] * 3, [p2] * 3, 2.0, 0.999, 20000, 3.0, seeds=[1, 2, 3]); sa = pd.last_stats()
    Hb, kb = pd.findHomographyBatch([p1] * 3, [p2] * 3, 2.0, 0.999, 20000, 3.0, seeds=[1, 2, 3], flags=_lib.FLAG_NO_HJOB); sb = pd.last_stats()
    assert np.array_equal(np.asarray(Ha), np.asarray(Hb)) and all(np.array_equal(np.asarray(a), np.asarray(b)) for a, b in zip(ka, kb)) and _key(sa) == _key(sb)


def test_a_hand_over_time_out_is_discarded_and_rerun_not_returned(oracle_port):
    """Fault injection (wait limit 0): every data wait of the stream mode fails, owners give up on their producers.  The host-pointer entry points
    must still return the oracle's results (the affected pairs run again without producers: stats bit 11); the asynchronous entry
    point must not return invalid numbers as a success: affected pairs come back as zero model + zero mask + stats bit 10."""
    import torch
    A, B, seeds = _f_batch(8)
    ora = [oracle_port.find_fundamental(A[p], B[p], 0.5, 0.9999, 30000, seed=seeds[p]) for p in range(len(A))]
    prev = _lib.set_wait_ticks(0)
    try:
        flags = _lib.FLAG_STREAM_ON | _lib.FLAG_STREAM_TEST(2)
        F, m = pd.findFundamentalMatrixBatch(A, B, max_iters=30000, seeds=seeds, flags=flags); st = pd.last_stats()
        assert sum(s_["rerun"] for s_ in st) >= 1, "the tiny limit must have tripped at least one wait"
        for p, (Fo, mo, so) in enumerate(ora):
            assert (st[p]["samples"], st[p]["lo_runs"], st[p]["I"]) == (so["samples"], so["lo_runs"], so["I"]), p
            assert np.array_equal(np.asarray(m[p]), mo), p
            assert np.linalg.norm(np.asarray(F[p]).ravel() - Fo.ravel()) <= 1e-9 * np.linalg.norm(Fo), p
            assert st[p]["discarded"] == 0
        # the asynchronous device-pointer entry point: nothing to retry with, the failure must be visible in the outputs
        dev = torch.device("cuda", 0)
        offs = np.zeros(len(A) + 1, np.int64); offs[1:] = np.cumsum([a.shape[0] for a in A])
        d_a = torch.from_numpy(np.concatenate(A)).to(dev); d_b = torch.from_numpy(np.concatenate(B)).to(dev); d_off = torch.from_numpy(offs).to(dev)
        d_seeds = torch.tensor(seeds, dtype=torch.int32, device=dev)
        d_F = torch.full((len(A), 9), 7.0, dtype=torch.float64, device=dev); d_mask = torch.full((int(offs[-1]),), 3, dtype=torch.uint8, device=dev)
        d_st = torch.zeros((len(A), 16), dtype=torch.int32, device=dev)
        prm = _lib.make_params(0.5, 0.9999, 30000, 0, True, 0.0, True, flags)
        stream = torch.cuda.current_stream(dev)
        _lib.check(_lib.lib().mi_degensac_find_fundamental_batch_dev(d_a.data_ptr(), d_b.data_ptr(), d_off.data_ptr(), offs.ctypes.data_as(C.POINTER(C.c_int64)),
                   len(A), 2, C.byref(prm), d_seeds.data_ptr(), 0, C.c_void_p(stream.cuda_stream), d_F.data_ptr(), d_mask.data_ptr(), d_st.data_ptr()))
        torch.cuda.synchronize(dev)
        hs = d_st.cpu().numpy(); hF = d_F.cpu().numpy(); hm = d_mask.cpu().numpy()
        n_disc = 0
        for p, (Fo, mo, so) in enumerate(ora):
            if (hs[p, 15] >> 10) & 1:
                n_disc += 1
                assert not hF[p].any() and not hm[offs[p]:offs[p + 1]].any() and hs[p, 3] == 0, p
            else:
                assert np.array_equal(hm[offs[p]:offs[p + 1]].astype(bool), mo) and hs[p, 0] == so["samples"], p
        assert n_disc >= 1
    finally:
        _lib.set_wait_ticks(prev)
    F, m = pd.findFundamentalMatrixBatch(A, B, max_iters=30000, seeds=seeds, flags=_lib.FLAG_STREAM_ON | _lib.FLAG_STREAM_TEST(2)); st = pd.last_stats()
    assert sum(s_["rerun"] + s_["discarded"] for s_ in st) == 0 and sum(s_["streamed"] for s_ in st) >= 1      # with the default limit nothing trips


def test_environment_tuning_word_is_masked_not_rejected():
    """MI_DEGENSAC_TUNING is process-wide and meets both kinds of call: a homography-only bit in it must not fail fundamental-matrix
    calls (and the other way round); the same bit in params.tuning of the wrong call is still an error."""
    code = ("import sys; sys.path.insert(0, %r)\n"
            "import numpy as np, pydegensac_amd as pd\n"
            "from pydegensac_amd import synthetic as syn, _lib\n"
            "a, b, _, _ = syn.two_view_fundamental(300, 0.5, 0.1, seed=1)\n"
            "F, m = pd.findFundamentalMatrix(a, b, 0.5, 0.9999, 2000, seed=3)\n"
            "h1, h2, _, _ = syn.homography_pairs(300, 0.5, 0.5, seed=1)\n"
            "H, k = pd.findHomography(h1, h2, 2.0, 0.999, 2000, seed=3)\n"
            "assert np.asarray(m).sum() > 50 and np.asarray(k).sum() > 50\n"
            "try:\n"
            "    pd.findFundamentalMatrix_(a, b, tuning=_lib.TUNE_H_SERIAL_LO, seed=1); raise SystemExit(3)\n"
            "except ValueError:\n"
            "    pass\n"
            "print('ok')\n") % ROOT
    env = dict(os.environ, MI_DEGENSAC_TUNING=str(_lib.TUNE_H_SERIAL_LO | _lib.TUNE_F_SERIAL_REPS | _lib.TUNE_COOP_ALL_PASSES | _lib.TUNE_LONG_SHIFT(2)))
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "ok" in out.stdout, out.stderr[-800:]


def test_homography_screen_on_the_device_is_a_superset(oracle_port):
    """mi_degensac_screen_counts_h: the gfx950 build of dg_HDs_maybe_below (per point) and of the four-models-per-sweep count
    (dg_h_screen4) against the reference's HDs (Htools.c:161-200, through the oracle): every point below 9/4 th is a candidate,
    and the swept count equals the number of candidates."""
    L = _lib.lib(); rng = np.random.default_rng(5); n = 3000; checked = 0
    for trial in range(24):
        scale = [1.0, 1e-3, 1e3, 1e-6][trial % 4]
        M = rng.normal(size=(7, 9)) * scale
        M[1, 6:9] = 0; M[2] = np.eye(3).ravel() + 1e-9 * rng.normal(size=9); M[3, 2] = M[3, 5] = 0; M[3, 8] = 1e-12
        M[4] = syn.H_1_6.T.ravel() if hasattr(syn, "H_1_6") else M[4]
        p1 = rng.uniform(-2000, 2000, size=(n, 2)); p2 = rng.uniform(-2000, 2000, size=(n, 2))
        if trial % 3 == 0: p2 = p1 + rng.normal(size=(n, 2))
        if trial % 7 == 0: p1[:, 1] = 2 * p1[:, 0] + 1
        p1[:5] = 0; p2[:5] = 0
        u = np.ones((n, 6)); u[:, 0:2] = p1; u[:, 3:5] = p2
        for th in (0.25, 4.0, 400.0):
            cnt = np.zeros(7, np.uint32); cand = np.zeros((7, n), np.uint8)
            _lib.check(L.mi_degensac_screen_counts_h(_lib.dptr(p1), _lib.dptr(p2), n, 2, _lib.dptr(M), 7, th, 0,
                                                     cnt.ctypes.data_as(C.POINTER(C.c_uint32)), cand.ctypes.data_as(C.POINTER(C.c_uint8))))
            for k in range(7):
                d = np.zeros(n)
                oracle_port.lib().dg_oracle_HDs(oracle_port.dp(u), oracle_port.dp(M[k].copy()), oracle_port.dp(d), n)
                inl = d < th * 9 / 4
                assert not np.any(inl & (cand[k] == 0)), (trial, th, k, int(np.sum(inl & (cand[k] == 0))))
                assert int(cnt[k]) == int(cand[k].sum()), (trial, th, k)
                checked += int(inl.sum())
    assert checked > 10000


def test_h2el_no_model_means_no_inliers():
    """findHomography's convention (utils.py:104-107) for ransacH2el's user-facing return: a zero H comes with an all-false mask"""
    u10, _ = syn.ellipse_pairs(300, 0.0, 1.0, 301, 0.05)
    H, m = pd.ransacH2el(u10, 4.0, 0.99, 500, seed=1)
    if not np.asarray(H).any():
        assert not np.asarray(m).any()
    Hr, mr = pd.ransacH2el(u10, 4.0, 0.99, 500, seed=1, raw=True)          # the driver's own arrays are returned untouched
    assert np.asarray(mr).shape == np.asarray(m).shape


def test_laf_rows_in_batches_set_aside_and_streamed(oracle_port):
    """[N, 6] rows with the LAF check through the BATCH paths: a ragged batch on a capped grid with pairs set aside and resumed, the
    same batch with producers (stream mode) forced on, and over a device list — every pair against the restatement."""
    A, B = [], []
    for i in range(14):
        n = [2000, 600, 1500, 300, 1000][i % 5]
        p1, p2, _, _ = syn.two_view_fundamental(n, 0.4, 0.1, seed=700 + i, plane_fraction=0.7 if i % 3 == 1 else 0.0, laf=True, laf_bad=0.5 if i % 2 else 0.25)
        A.append(p1); B.append(p2)
    seeds = [5 + 11 * i for i in range(14)]
    ora = [oracle_port.find_fundamental(A[p], B[p], 0.5, 0.9999, 20000, 0, True, 2.0, True, seed=seeds[p]) for p in range(14)]
    assert sum(o[2]["rejected"] for o in ora) > 0
    runs = [("set aside on 4 workgroups", dict(tuning=_lib.TUNE_THROUGHPUT | _lib.TUNE_GRID_CAP(4) | _lib.TUNE_SET_ASIDE(2))),
            ("producers", dict(flags=_lib.FLAG_STREAM_ON | _lib.FLAG_STREAM_TEST(2))),
            ("device list", dict(devices=[0, 0, 0]))]
    for tag, kw in runs:
        F, m = pd.findFundamentalMatrixBatch(A, B, 0.5, 0.9999, 20000, 2.0, seeds=seeds, **kw); st = pd.last_stats()
        for p, (Fo, mo, so) in enumerate(ora):
            assert (st[p]["samples"], st[p]["lo_runs"], st[p]["rejected"], st[p]["I"]) == (so["samples"], so["lo_runs"], so["rejected"], so["I"]), (tag, p)
            assert np.array_equal(np.asarray(m[p]), mo), (tag, p)
            assert np.linalg.norm(np.asarray(F[p]).ravel() - Fo.ravel()) <= 1e-9 * np.linalg.norm(Fo), (tag, p)
        if tag.startswith("set aside"):
            assert sum(s_["set_aside"] for s_ in st) >= 3


def test_bound_that_falls_inside_a_chunk(oracle_port):
    """exp_ranF.c:1478-1480: after a plane-and-parallax completion `maxS.J` becomes the MSAC sum of THAT model and may fall below the
    bound the current chunk was screened against; models of later samples of the chunk between the two must be looked at again.  The
    case the fuzz sweep found through the LAF-rejection counter (sweep seed 21, case 1771: 3 rejections counted against the
    reference's 10, results equal): every workgroup size and placement, own sample stream and producer ring."""
    p1, p2, _, _ = syn.two_view_fundamental(150, 0.7889425968606831, 0.1, seed=1771, plane_fraction=0.9, laf=True, laf_bad=0.5, laf_sigma=0.05)
    seed = 1847496781
    Fo, mo, so = oracle_port.find_fundamental(p1, p2, 2.0, 0.9999, 20000, 1, False, 2.0, True, seed=seed)
    assert so["rejected"] == 10 and so["degen"] == 3
    for variant in (1, 2, 3):
        for mode in (1, 2, 3):
            for flags in (_lib.FLAG_NO_STREAM, _lib.FLAG_STREAM_ON | _lib.FLAG_STREAM_TEST(2)):
                F, m = pd.findFundamentalMatrix_(p1, p2, 2.0, 0.9999, 20000, 1, False, 2.0, True, seed=seed, flags=flags, tuning=variant | (mode << 2)); st = pd.last_stats()
                assert (st["samples"], st["lo_runs"], st["degen"], st["rejected"], st["I"]) == (so["samples"], so["lo_runs"], so["degen"], so["rejected"], so["I"]), (variant, mode, flags, st)
                assert np.array_equal(np.asarray(m), mo) and np.linalg.norm(np.asarray(F).ravel() - Fo.ravel()) <= 1e-9 * np.linalg.norm(Fo)
    # the same scene without the LAF check: the candidates the LAF check turned down are now ACCEPTED by the reference
    Fo, mo, so = oracle_port.find_fundamental(p1[:, :2].copy(), p2[:, :2].copy(), 2.0, 0.9999, 20000, 1, False, 0.0, True, seed=seed)
    for variant in (1, 2, 3):
        for flags in (_lib.FLAG_NO_STREAM, _lib.FLAG_STREAM_ON | _lib.FLAG_STREAM_TEST(2)):
            F, m = pd.findFundamentalMatrix_(p1[:, :2].copy(), p2[:, :2].copy(), 2.0, 0.9999, 20000, 1, False, 0.0, True, seed=seed, flags=flags, tuning=variant); st = pd.last_stats()
            assert (st["samples"], st["lo_runs"], st["degen"], st["I"]) == (so["samples"], so["lo_runs"], so["degen"], so["I"]), (variant, flags)
            assert np.array_equal(np.asarray(m), mo)


@pytest.mark.gpu
def test_run_after_the_loop_when_no_sample_ever_beat_the_running_best(oracle_port):
    """exp_ranH.c:759-862 / ranH2el.c:163-187: with no local optimisation so far the drivers run one after the loop from errs[4] —
    which still is the buffer it started as (errs[4] = errs[3], exp_ranH.c:531) when every sample was rejected or none had an inlier.
    The reference reads its uninitialised allocation there; the oracle and the device take a zero-filled one (every point within the
    threshold: the least squares runs over ALL points).  Found by `tools/gpu_fuzz.py edges` (sample budgets of 1-7): the device used
    to score the all-zero model instead (NaN residuals: no point, or every point, depending on the metric)."""
    hit = 0
    for seed in range(60):
        mi = 1 + seed % 3; et = seed % 5
        p1, p2, _, _ = syn.homography_pairs(64, 0.5, 0.1, seed=1400 + seed)
        for tn in (1, 2, 3):
            Hg, mg = pd.findHomography_(p1, p2, 10.0, 0.9, mi, et, True, 0.0, seed=345625506 + seed, tuning=tn); sg = pd.last_stats()
            Ho, mo, so = oracle_port.find_homography(p1, p2, 10.0, 0.9, mi, et, True, 0.0, seed=345625506 + seed)
            assert (sg["samples"], sg["lo_runs"], sg["rejected"], sg["I"], sg["models"], sg["best_sample"]) == \
                   (so["samples"], so["lo_runs"], so["rejected"], so["I"], so["models"], so["best_sample"]), (seed, tn)
            assert np.array_equal(np.asarray(mg, bool), np.asarray(mo, bool)) and np.allclose(np.asarray(Hg).ravel(), np.asarray(Ho).ravel(), rtol=1e-9, atol=0), (seed, tn)
        hit += so["rejected"] == so["samples"]                # every sample turned down before scoring: errs[4] never written
    assert hit >= 3, hit
    # ransacH2el: a threshold no sample meets
    u10 = syn.ellipse_pairs(100, 0.4, 1.0, 5)[0]
    for mi in (1, 3, 60):
        H, m = pd.ransacH2el_batch([u10], 1e-6, 0.99, mi, True, 0, seeds=[11], raw=True); st = pd.last_stats()[0]
        Ho, mo, so = oracle_port.ransacH2el(u10, 1e-6, 0.99, mi, True, 0, 11)
        assert (st["samples"], st["lo_runs"], st["I"], st["models"]) == (so["samples"], so["lo_runs"], so["I"], so["models"]), mi
        assert np.allclose(np.asarray(H[0]).ravel(), np.asarray(Ho).ravel(), rtol=1e-9, atol=0)


def test_errs4_buffer_written_again_inside_the_sample_that_set_it(oracle_port):
    """Three models of ONE sample past sample 50 (exp_ranF.c:1365-1495): the first beats the best sample score, is accepted as the best
    model (its residual buffer becomes errs[3]) and sets errs[4]; the third is accepted too — errs[3]'s old buffer becomes ITS errs[i]
    (:1409-1410) — and is degenerate, so the plane-and-parallax model's residuals are written there (:1463-1466): the local
    optimisation at the end of the sample reads THOSE through errs[4].  The device used to take the first model's (the bookkeeping of
    errs[4]'s buffer was off past sample 50): `tools/gpu_fuzz.py 6000 202`, case 3465 — 208 samples / 41 inliers against the
    reference's 209 / 42."""
    p1, p2, _, _ = syn.two_view_fundamental(64, 0.6202637445729403, 0.5, seed=3465, plane_fraction=0.9)
    Fo, mo, so = oracle_port.find_fundamental(p1, p2, 2.0, 0.9999, 3000, 0, True, 0.0, True, seed=106905411)
    assert (so["samples"], so["lo_runs"], so["I"], so["degen"]) == (209, 3, 42, 6), so          # = the unmodified reference's run
    for tn in (1 | (3 << 2), 1 | (1 << 2), 2 | (3 << 2), 3 | (1 << 2)):
        for fl in (0, _lib.FLAG_STREAM_ON | _lib.FLAG_STREAM_TEST(2)):
            Fg, mg = pd.findFundamentalMatrix_(p1, p2, 2.0, 0.9999, 3000, 0, True, 0.0, True, seed=106905411, tuning=tn, flags=fl); sg = pd.last_stats()
            assert [sg[k] for k in ("samples", "lo_runs", "I", "models", "best_sample", "degen")] == [so[k] for k in ("samples", "lo_runs", "I", "models", "best_sample", "degen")], (tn, fl)
            assert np.array_equal(np.asarray(mg, bool), mo) and np.allclose(np.asarray(Fg).ravel(), np.asarray(Fo).ravel(), rtol=1e-9, atol=0), (tn, fl)
