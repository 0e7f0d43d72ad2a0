"""GPU: the unit-level entry points of include/mi_degensac.h against the oracle's building blocks, and the
boundary's concurrency contract (threads x streams on one device)."""
import ctypes as C
import threading

import numpy as np
import pytest

import pydegensac_amd as pd
from pydegensac_amd import _lib, synthetic as syn
from tests import golden_util as gu

pytestmark = pytest.mark.gpu


def test_solve7_matches_oracle_nullspace_cubic_and_orientation(oracle_port):
    """mi_degensac_solve7 = the per-lane 7-point solver of the main kernel: null space (utools.c:97-167), cubic
    (Ftools.c:39-81), real roots (Ftools.c:251-298), oriented epipolar test (Ftools.c:463-494), exp_ranF.c:1351-1372."""
    L = _lib.lib(); P = oracle_port.lib(); dp = oracle_port.dp; ip = oracle_port.ip
    n = 1500
    p1, p2, lab, _ = syn.two_view_fundamental(n, 0.5, 0.1, seed=4)
    u = np.ones((n, 6)); u[:, 0:2] = p1; u[:, 3:5] = p2
    rng = np.random.default_rng(1)
    S = 2000
    samples = np.stack([rng.choice(n, 7, replace=False) for _ in range(S)]).astype(np.int32)
    inl = np.flatnonzero(lab)
    samples[:300] = np.stack([rng.choice(inl, 7, replace=False) for _ in range(300)])      # all-inlier samples: valid models
    samples[300, 1] = samples[300, 0]                                                       # a rank-deficient sample
    nsol = np.zeros(S, np.int32); ridx = np.zeros((S, 3), np.int32); models = np.zeros((S, 27))
    _lib.check(L.mi_degensac_solve7(_lib.dptr(p1), _lib.dptr(p2), n, 2, samples.ctypes.data_as(C.POINTER(C.c_int32)), S, 0,
                                    nsol.ctypes.data_as(C.POINTER(C.c_int32)), ridx.ctypes.data_as(C.POINTER(C.c_int32)), _lib.dptr(models)))
    n_models = 0; n_exact = 0; n_same_roots = 0; polys = {}; want_all = {}
    for t in range(S):
        A = np.zeros(81)
        for i, q in enumerate(samples[t]):                                   # rows in draw order (rtools.c:74-92)
            A[9 * i:9 * i + 9] = np.outer(u[q, 3:6], u[q, 0:3]).ravel()
        ns = np.zeros(81)
        dim = P.dg_oracle_nullspace(dp(A), dp(ns), 9)
        if dim != 2:
            assert nsol[t] == -1, t
            continue
        f1 = ns[0:9].copy(); f2 = ns[9:18].copy(); poly = np.zeros(4); roots = np.zeros(3)
        P.dg_oracle_slcm(dp(f1), dp(f2), dp(poly))                           # f2 := f1 - f2 (Ftools.c:70)
        nr = P.dg_oracle_rroots3(dp(poly), dp(roots))
        samidx = np.ascontiguousarray(samples[t][::-1])                      # pool tail = reverse draw order (rtools.c:17-20)
        want = []
        for i in range(nr):
            f = f1 * roots[i] + f2 * (1 - roots[i])                          # exp_ranF.c:1365-1368
            if P.dg_oracle_all_ori_valid(dp(np.ascontiguousarray(f)), dp(u), ip(samidx), 7):
                want.append((i, f))
        assert nsol[t] == len(want), (t, nsol[t], len(want))
        polys[t] = (poly.copy(), nr, roots.copy()); want_all[t] = want
        for k, (i, f) in enumerate(want):
            # rroots3 goes through pow / acos / cos (Ftools.c:251-298): the device takes their correctly rounded values
            # (dg_crmath.h), the host's libm returns those in all but ~0.1-0.2 % of its calls; a root that differs in its last bit
            # is amplified by cancellation in the model entries.  Everything else is the same IEEE operation sequence
            got = models[t, 9 * k:9 * k + 9]
            assert ridx[t, k] == i and np.abs(got - f).max() <= 1e-10 * np.abs(f).max(), (t, k)
            n_exact += int(np.array_equal(got, f))
        n_models += len(want)
    assert n_models > 200 and (nsol == -1).any() and n_exact > 0.98 * n_models, (n_models, n_exact)
    # where do the inexact ones come from?  The cubic's roots on the device (mi_degensac_mat3 op 4 = the solver's own rroots3)
    # against the host's: wherever a root has the same bits, so has its model -- pow / acos / cos are the only operations of
    # the solver that are not the reference's IEEE sequence, and with their correctly rounded values (dg_crmath.h) more than
    # 99 % of the roots carry the host's bits (96.6 % with the device library's own functions)
    ts = sorted(polys); PO = np.ascontiguousarray(np.stack([polys[t][0] for t in ts]))
    R = np.zeros((len(ts), 3)); nrd = np.zeros(len(ts), np.int32)
    _lib.check(L.mi_degensac_mat3(4, _lib.dptr(PO), len(ts), 0, _lib.dptr(R), nrd.ctypes.data_as(C.POINTER(C.c_int32))))
    n_roots = 0; n_roots_same = 0; worst = 0.0
    for a, t in enumerate(ts):
        poly, nr, roots = polys[t]
        assert nrd[a] == nr, (t, nrd[a], nr)
        scale = max(1.0, np.abs(roots[:nr]).max(), abs(poly[1] / poly[0]) / 3)
        for i in range(nr):
            n_roots += 1; n_roots_same += int(R[a, i] == roots[i]); worst = max(worst, abs(R[a, i] - roots[i]) / scale)
        for k, (i, f) in enumerate(want_all[t]):
            if R[a, i] == roots[i]:
                n_same_roots += 1
                assert np.array_equal(models[t, 9 * k:9 * k + 9], f), (t, k)
    assert n_same_roots > 100 and n_roots_same > 0.99 * n_roots and worst < 1e-14, (n_same_roots, n_roots_same, n_roots, worst)
    print(f"rroots3: {n_roots_same}/{n_roots} roots with the host's bits, worst difference {worst:.1e} of the scale; "
          f"{n_exact}/{n_models} models bit-equal, all {n_same_roots} with a bit-equal root among them")


def test_score_models_symmetric_h_metrics_bit_exact(oracle_port):
    """mi_degensac_score_models kinds 10..14 (H Sampson and the four symmetric transfer errors, Htools.c:161-370):
    residuals bit-exact, I exact, J = the reference's sequential MSAC sum"""
    L = _lib.lib(); P = oracle_port.lib(); dp = oracle_port.dp; ip = oracle_port.ip
    n = 2500
    p1, p2, lab, Hgt = syn.homography_pairs(n=n, inlier_ratio=0.4, sigma=0.5, seed=8)
    u = np.ones((n, 6)); u[:, 0:2] = p1; u[:, 3:5] = p2
    rng = np.random.default_rng(2)
    Hc = np.linalg.inv(Hgt).T.ravel()
    models = np.stack([Hc] + [Hc * (1 + 0.003 * rng.normal(size=9)) for _ in range(7)] + [rng.normal(size=9) for _ in range(4)]).copy()
    M = len(models)
    for kind, th in [(0, 4.0), (1, 4.0), (2, 2.0), (3, 4.0), (4, 2.0)]:
        I = np.zeros(M, np.uint32); J = np.zeros(M); res = np.zeros((M, n))
        _lib.check(L.mi_degensac_score_models(_lib.dptr(p1), _lib.dptr(p2), n, 2, _lib.dptr(models), M, 10 + kind, th, 0,
                                              I.ctypes.data_as(C.POINTER(C.c_uint32)), _lib.dptr(J), _lib.dptr(res)))
        for k in range(M):
            d = np.zeros(n)
            P.dg_oracle_HDS_full(kind, dp(u), dp(models[k].copy()), dp(d), n)
            lst = np.zeros(n, np.int32)
            S = P.dg_oracle_inlidxs(dp(d), n, C.c_double(th), ip(lst))
            assert np.array_equal(d, res[k], equal_nan=True), (kind, k)
            assert S.I == I[k] and (S.J == J[k] or (np.isnan(S.J) and np.isnan(J[k]))), (kind, k, S.J, J[k])
        assert I[0] > 0.1 * n


def test_two_threads_two_streams_equal_serial_runs(oracle_port):
    """thread-safe boundary.  Two host threads, each driving two streams of the same device through the
    asynchronous *_dev entry points (ragged batches of different sizes so the per-stream scratch differs), plus the
    host-pointer API from both threads at once: every result equals the serial run bit for bit."""
    import torch
    from pydegensac_amd import tensor_api
    dev = torch.device("cuda", 0)
    jobs = []
    for j in range(4):
        sizes = [300 + 137 * ((j + k) % 5) for k in range(6 + 3 * j)]
        A = []; B = []
        for i, n in enumerate(sizes):
            p1, p2, _, _ = syn.two_view_fundamental(n, 0.5, 0.1, seed=100 * j + i); A.append(p1); B.append(p2)
        jobs.append((sizes, torch.from_numpy(np.concatenate(A)).to(dev), torch.from_numpy(np.concatenate(B)).to(dev),
                     [17 * j + i + 1 for i in range(len(sizes))], A, B))
    serial = []
    for sizes, a, b, seeds, _, _ in jobs:
        F, m, st, _ = tensor_api.find_fundamental_batch_tensors(a, b, sizes, max_iters=5000, seeds=seeds)
        torch.cuda.synchronize()
        serial.append((F.cpu().numpy().copy(), m.cpu().numpy().copy()))
    # spot check of the serial run itself against the oracle
    Fo, mo, _ = oracle_port.find_fundamental(jobs[0][4][0], jobs[0][5][0], 0.5, 0.9999, 5000, seed=jobs[0][3][0])
    assert np.array_equal(serial[0][1][:jobs[0][0][0]], mo) and gu.rel(serial[0][0][0], Fo) < 1e-6
    hostF, hostm = pd.findFundamentalMatrixBatch(jobs[1][4], jobs[1][5], max_iters=5000, seeds=jobs[1][3])
    results = {}; errors = []

    def worker(tid):
        try:
            torch.cuda.set_device(0)
            streams = [torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)]
            for rep in range(3):
                outs = []
                for k in range(2):
                    sizes, a, b, seeds, _, _ = jobs[2 * tid + k]
                    with torch.cuda.stream(streams[k]):
                        outs.append(tensor_api.find_fundamental_batch_tensors(a, b, sizes, max_iters=5000, seeds=seeds))
                hf, hm = pd.findFundamentalMatrixBatch(jobs[1][4], jobs[1][5], max_iters=5000, seeds=jobs[1][3])   # host API, same time
                for k in range(2):
                    streams[k].synchronize()
                    results[(tid, rep, k)] = (outs[k][0].cpu().numpy().copy(), outs[k][1].cpu().numpy().copy())
                results[(tid, rep, "host")] = (np.asarray(hf).copy(), [np.asarray(x).copy() for x in hm])
        except Exception as e:                                  # noqa: BLE001
            errors.append(repr(e))

    th = [threading.Thread(target=worker, args=(t,)) for t in range(2)]
    [t.start() for t in th]; [t.join() for t in th]
    assert not errors, errors
    for tid in range(2):
        for rep in range(3):
            for k in range(2):
                F, m = results[(tid, rep, k)]
                assert np.array_equal(F, serial[2 * tid + k][0]) and np.array_equal(m, serial[2 * tid + k][1]), (tid, rep, k)
            hf, hm = results[(tid, rep, "host")]
            assert np.array_equal(hf, np.asarray(hostF)) and all(np.array_equal(x, np.asarray(y)) for x, y in zip(hm, hostm))


def test_explicit_context_and_device_is_restored():
    """mi_degensac_ctx_*: an explicit context gives the same bits as the thread's implicit one; the calling thread's
    current HIP device is untouched (here: checked through torch's notion of the current device)."""
    import torch
    L = _lib.lib()
    p1, p2, _, _ = syn.two_view_fundamental(600, 0.5, 0.1, seed=9)
    F0, m0 = pd.findFundamentalMatrix(p1, p2, max_iters=4000, seed=21)
    ctx = C.c_void_p()
    _lib.check(L.mi_degensac_ctx_create(0, C.byref(ctx)))
    try:
        assert L.mi_degensac_ctx_stream(ctx) is not None
        prm = _lib.make_params(0.5, 0.9999, 4000, 0, True, 0.0, True)
        off = np.array([0, 600], np.int64); sd = np.array([21], np.uint32)
        F = np.zeros(9); mk = np.zeros(600, np.uint8); st = np.zeros(16, np.int32)
        for _ in range(2):                                           # second call reuses the staging buffers
            _lib.check(L.mi_degensac_ctx_find_fundamental_batch(ctx, _lib.dptr(p1), _lib.dptr(p2), off.ctypes.data_as(C.POINTER(C.c_int64)), 1, 2,
                                                               C.byref(prm), sd.ctypes.data_as(C.POINTER(C.c_uint32)), _lib.dptr(F),
                                                               mk.ctypes.data_as(C.POINTER(C.c_uint8)), st.ctypes.data_as(C.POINTER(C.c_int32))))
            assert np.array_equal(F.reshape(3, 3), F0) and np.array_equal(mk.astype(bool), np.asarray(m0))
    finally:
        L.mi_degensac_ctx_destroy(ctx)
    assert torch.cuda.current_device() == 0
    assert L.mi_degensac_release_scratch(0, None) == 0


def test_mat3_device_routines_equal_reference(oracle_ref):
    """mi_degensac_mat3: the lane-level 3x3 inverse / right singular vectors / Hdetect as compiled for gfx950, against
    the unmodified reference's minv / svduv / Hdetect (oracle/_ref) on 10^4 random inputs each: bit for bit"""
    from tests.test_mat3_cpu import _mats
    L = _lib.lib(); R = oracle_ref.lib()
    dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
    rng = np.random.default_rng(21)
    A = np.stack(list(_mats(rng, 10000)))
    out = np.zeros((len(A), 9)); flag = np.zeros(len(A), np.int32)
    _lib.check(L.mi_degensac_mat3(0, dp(A), len(A), 0, dp(out), flag.ctypes.data_as(C.POINTER(C.c_int32))))
    for a, o, f in zip(A, out, flag):
        y = a.copy(); r = R.minv(dp(y), 3)
        assert (r != 0) == (f != 0) and np.array_equal(o, y.ravel(), equal_nan=True)
    out = np.zeros((len(A), 12))
    _lib.check(L.mi_degensac_mat3(1, dp(A), len(A), 0, dp(out), flag.ctypes.data_as(C.POINTER(C.c_int32))))
    for a, o in zip(A, out):
        y = a.copy(); d = np.zeros(3); u = np.zeros(9); v = np.zeros(9)
        R.svduv(dp(d), dp(y), dp(u), 3, dp(v), 3)
        assert np.array_equal(o[:9], v, equal_nan=True) and np.array_equal(o[9:], d, equal_nan=True)
    trip = np.array([[0, 1, 2], [3, 4, 5], [0, 1, 6], [3, 4, 6], [2, 5, 6]], np.uint8)
    N = 10000; inp = np.zeros((N, 40)); want = np.zeros((N, 9))
    for t in range(N):
        F = rng.normal(size=(3, 3))
        if t % 3:
            u, s, vt = np.linalg.svd(F); s[2] = 0; F = (u * s) @ vt
        pts = rng.uniform(-500, 500, size=(7, 4))
        if t % 7 == 0:
            pts[2] = pts[0] + (pts[1] - pts[0]) * 0.3
        ids = np.ascontiguousarray(trip[t % 5])
        inp[t, :9] = F.ravel(); inp[t, 9:37] = pts.ravel(); inp[t, 37:] = ids
        u7 = np.ones((7, 6)); u7[:, 0:2] = pts[:, 0:2]; u7[:, 3:5] = pts[:, 2:4]
        R.Hdetect(dp(np.ascontiguousarray(F).copy()), dp(u7), ids.ctypes.data_as(C.POINTER(C.c_ubyte)), dp(want[t]))
    out = np.zeros((N, 9)); flag = np.zeros(N, np.int32)
    _lib.check(L.mi_degensac_mat3(2, dp(inp), N, 0, dp(out), flag.ctypes.data_as(C.POINTER(C.c_int32))))
    assert np.array_equal(out, want, equal_nan=True), np.flatnonzero((out != want).any(axis=1))[:10]


def test_wave_eigensolver_equals_oracle_dsyev():
    """mi_degensac_mat3 op 3: the wave eigen-solver (dsytd2 + dorg2l + the replicated-register dsteqr of dg_steqr9.h) on
    symmetric 9x9 matrices — Gram matrices of normalised correspondence rows like the estimator's, random symmetric
    ones over many scales, rank-deficient ones — against the CPU oracle's dsyev restatement: the smallest eigenvalue
    and its eigenvector (all the estimator reads) bit for bit, and the whole spectrum as a set."""
    from oracle import port
    L = _lib.lib(); O = port.lib()
    dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
    rng = np.random.default_rng(33)
    N = 4000; A = np.zeros((N, 9, 9))
    for t in range(N):
        k = t % 5
        if k == 0:
            m = rng.normal(size=(rng.integers(8, 60), 9)); a = m.T @ m
        elif k == 1:
            x1 = rng.normal(size=(14, 2)) * np.sqrt(2) / 2; x2 = x1 + rng.normal(size=(14, 2)) * 0.05
            m = np.stack([np.r_[b[0] * np.r_[a_, 1.0], b[1] * np.r_[a_, 1.0], np.r_[a_, 1.0]] for a_, b in zip(x1, x2)]); a = m.T @ m
        elif k == 2:
            a = rng.normal(size=(9, 9)) * 10.0 ** rng.integers(-5, 6); a = a + a.T
        elif k == 3:
            m = rng.normal(size=(6, 9)); a = m.T @ m                       # rank 6
        else:
            a = np.diag(rng.normal(size=9)) + 1e-9 * rng.normal(size=(9, 9)); a = (a + a.T) / 2
        A[t] = (a + a.T) / 2
    out = np.zeros((N, 90)); flag = np.zeros(N, np.int32)
    _lib.check(L.mi_degensac_mat3(3, dp(A), N, 0, dp(out), flag.ctypes.data_as(C.POINTER(C.c_int32))))
    bad = []
    for t in range(N):
        a = A[t].copy(); w = np.zeros(9)
        info = O.dg_oracle_eig_sym(dp(a), dp(w), 9)
        ok = info == flag[t] and out[t, 0] == w[0] and np.array_equal(out[t, 9:18], a.ravel()[:9]) \
            and np.array_equal(np.sort(out[t, :9]), w)
        if not ok:
            bad.append(t)
    assert not bad, (len(bad), bad[:10], flag[bad[:10]])


def test_two_eigenproblems_per_wave_equal_one_per_wave():
    """mi_degensac_mat3 op 5 (dg_eig2.h: problem 2t in lanes 0..31 of wave t, problem 2t + 1 in lanes 32..63, dsteqr's decisions per half-wave)
    against op 3 on the same matrices, bit for bit — every output number, the info word — in neighbouring and in random pairings (the two
    halves then take different QL / QR paths and sweep counts), odd counts included."""
    L = _lib.lib()
    dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
    rng = np.random.default_rng(35)
    N = 1501; A = np.zeros((N, 9, 9))
    for t in range(N):
        k = t % 5
        if k == 0:
            m = rng.normal(size=(rng.integers(8, 60), 9)); a = m.T @ m
        elif k == 1:
            x1 = rng.normal(size=(10, 2)) * np.sqrt(2) / 2; x2 = x1 + rng.normal(size=(10, 2)) * 0.05
            m = np.stack([np.r_[b[0] * np.r_[a_, 1.0], b[1] * np.r_[a_, 1.0], np.r_[a_, 1.0]] for a_, b in zip(x1, x2)]); a = m.T @ m
        elif k == 2:
            a = rng.normal(size=(9, 9)) * 10.0 ** rng.integers(-5, 6); a = a + a.T
        elif k == 3:
            m = rng.normal(size=(6, 9)); a = m.T @ m                       # rank 6
        else:
            a = np.diag(rng.normal(size=9)) + 1e-9 * rng.normal(size=(9, 9)); a = (a + a.T) / 2
        A[t] = (a + a.T) / 2

    def run(op, M):
        out = np.zeros((len(M), 90)); flag = np.zeros(len(M), np.int32)
        _lib.check(L.mi_degensac_mat3(op, dp(np.ascontiguousarray(M)), len(M), 0, dp(out), flag.ctypes.data_as(C.POINTER(C.c_int32))))
        return out, flag
    for trial in range(3):
        M = A if trial == 0 else A[rng.permutation(N)]
        o1, f1 = run(3, M); o2, f2 = run(5, M)
        assert np.array_equal(f1, f2), trial
        assert np.array_equal(o1, o2, equal_nan=True), (trial, np.flatnonzero((o1 != o2).any(axis=1))[:10])


def test_screening_counts_are_supersets_of_the_exact_band():
    """Level 1 (fp32, loosest denominator) and level 2 (fp64, own denominator) of the scoring phase's screens must never
    count fewer points than lie inside the 9/4 th band of the exact residuals, for random models, for models fitted to the
    data (many points ON the band edge region), and for adversarial ones: tiny and huge scales, large coordinates."""
    import ctypes as C
    L = _lib.lib()
    rng = np.random.default_rng(3)
    p1, p2, lab, Fgt = syn.two_view_fundamental(3000, 0.5, 0.3, seed=8)
    big1 = p1 * 37.0 + 5000.0; big2 = p2 * 37.0 - 9000.0                   # large coordinates: the fp32 bound must widen with them
    models = [rng.normal(size=9) for _ in range(300)]
    models += [Fgt.ravel() * s for s in (1.0, 1e-12, 1e12, -3.0)]
    models += [Fgt.ravel() + rng.normal(scale=10.0 ** -k, size=9) * np.abs(Fgt).max() for k in range(1, 9) for _ in range(12)]
    models += [rng.normal(size=9) * 10.0 ** rng.integers(-30, 30) for _ in range(100)]
    M = np.ascontiguousarray(np.array(models, dtype=np.float64)); nm = M.shape[0]
    for (a, b) in ((p1, p2), (big1, big2)):
        a = np.ascontiguousarray(a); b = np.ascontiguousarray(b); n = a.shape[0]
        for kind in (0, 1):
            for th in (0.25, 4.0):
                c1 = np.zeros(nm, np.uint32); c2 = np.zeros(nm, np.uint32)
                _lib.check(L.mi_degensac_screen_counts(_lib.dptr(a), _lib.dptr(b), n, 2, _lib.dptr(M), nm, kind, C.c_double(th), 0,
                                                       c1.ctypes.data_as(C.POINTER(C.c_uint32)), c2.ctypes.data_as(C.POINTER(C.c_uint32))))
                I = np.zeros(nm, np.uint32); J = np.zeros(nm); R = np.zeros((nm, n))
                _lib.check(L.mi_degensac_score_models(_lib.dptr(a), _lib.dptr(b), n, 2, _lib.dptr(M), nm, kind, C.c_double(th), 0,
                                                      I.ctypes.data_as(C.POINTER(C.c_uint32)), _lib.dptr(J), _lib.dptr(R)))
                exact = (R < th * 9 / 4).sum(axis=1)                            # the band of the MSAC gain, on the kernel's own residuals
                assert (c2 >= exact).all(), (kind, th, int(np.argmax(exact.astype(np.int64) - c2)))
                assert (c1 >= exact).all(), (kind, th, int(np.argmax(exact.astype(np.int64) - c1)))
                assert (exact > 0).any() and (c1[:300] < n).any()                 # the test has teeth: bands are hit, level 1 rejects



def test_wave_generator_equals_libc(oracle_port):
    """dg_srand_wave / dg_rand_skip / dg_rand_block (one wave: a 31 x 31 linear map for srand, three interleaved prefix sums for up to
    31 outputs at once) against the oracle's libc-faithful srand() / rand() (itself pinned on libc in tests/test_oracle_cpu.py):
    every block size, skips that wrap the ring, seeds 0, 1, 2^31 - 1, 2^31, 2^32 - 1."""
    import ctypes as C
    from pydegensac_amd import _lib
    L = _lib.lib()
    rng = np.random.default_rng(3)
    seeds = [0, 1, 2, 12345, 2**31 - 1, 2**31, 2**32 - 1] + [int(x) for x in rng.integers(1, 2**32 - 1, 12)]
    for i, seed in enumerate(seeds):
        for block in ([1, 2, 3, 8, 14, 16, 30, 31] if i < 4 else [int(rng.integers(1, 32))]):
            skip = int(rng.integers(0, 200)); count = 400
            got = np.zeros(count, np.int32)
            _lib.check(L.mi_degensac_rng_wave(seed, skip, block, count, 0, got.ctypes.data_as(C.POINTER(C.c_int32))))
            want = np.zeros(skip + count, np.int32)
            oracle_port.lib().dg_oracle_rand_stream(C.c_uint(seed), skip + count, oracle_port.ip(want))
            assert np.array_equal(got, want[skip:]), (seed, block, skip, int(np.argmax(got != want[skip:])))
