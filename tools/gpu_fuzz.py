"""Randomised GPU-vs-oracle sweep (F and H, all metrics, LAF on/off, degenerate scenes, both kernel variants and all
placement modes): masks and sample / LO counts must be identical, models equal to 1e-9.
    python tools/gpu_fuzz.py [n_cases] [seed]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import pydegensac_amd as pd
from pydegensac_amd import synthetic as syn
from oracle import port
LAF_STATS = [0, 0]          # F cases run with the LAF check on; of those, cases in which it turned a candidate down


class _Trace:
    """FUZZ_TRACE=1 (with FUZZ_ONLY=<case> and MI_DEGENSAC_LIB=tools/libmi_degensac_dev.so): the driver's checkpoints (tag, I, J) of
    one fundamental-matrix case on both sides -- tags 1, 2 around the least squares before an LO, 10-15 inside it, 30 a new best
    sample model (exp_ranF.c:1421), 31 the plane's support (:1430-1436), 32 after innerH, 33 after rFtH -- and the first that differs."""
    def __init__(self):
        import ctypes as C
        from pydegensac_amd import _lib
        self.C = C; self.L = _lib.lib(); self.P = port.lib(); self.o = []; self.cap = 20000
        self.CB = C.CFUNCTYPE(None, C.c_int, C.c_int, C.c_double); self.cb = self.CB(lambda t, i, j: self.o.append((t, i, j)))
        self.L.mi_degensac_debug_trace(self.cap, None)
    def gpu_done(self):
        C = self.C; buf = np.zeros(1 + 4 * self.cap, np.int32); self.L.mi_degensac_debug_trace(0, buf.ctypes.data_as(C.POINTER(C.c_int)))
        rec = buf[1:1 + 4 * buf[0]].reshape(-1, 4)
        self.g = [(int(r[0]), int(r[1]), float(np.array([(int(r[2]) & 0xffffffff) | (int(r[3]) << 32)], dtype=np.int64).view(np.float64)[0])) for r in rec]
        self.P.dg_oracle_set_trace2(self.cb)
    def report(self):
        self.P.dg_oracle_set_trace2(self.CB(0))
        keep = lambda tr: [x for x in tr if x[0] in (1, 2, 12, 30, 31, 32, 33)]
        o, g = keep(self.o), keep(self.g)
        for i, (a, b) in enumerate(zip(o, g)):
            if a != b:
                print(f"   trace: first difference at event {i} of {len(o)} / {len(g)}  (tag, I, J)")
                lo_ = 0 if os.environ.get("FUZZ_TRACE") == "all" else max(0, i - 3)
                for j in range(lo_, min(len(o), len(g), i + 4)): print("     oracle", o[j], " gpu", g[j])
                return
        print(f"   trace: {len(o)} / {len(g)} events, equal over the common prefix")


def run(N, rng_seed, verbose=True):
    """Returns (cases with different results, cases where only sample / LO counters differ)."""
    rng = np.random.default_rng(rng_seed)
    bad = 0; bad_res = 0
    if True:
        only = os.environ.get("FUZZ_ONLY")                      # debugging: run only these case numbers (the random stream still advances)
        only = set(int(x) for x in only.split(",")) if only else None
        for case in range(N):
            variant = int(rng.choice([512, 256, 128])); mode = int(rng.choice([0, 1, 2]))
            tn = {512: 1, 256: 2, 128: 3}[variant] | ((mode + 1) << 2)          # params.tuning: variant, placement (include/mi_degensac.h)
            if os.environ.get("FUZZ_TUNING"):                   # debugging: this kernel variant / placement instead of the drawn one
                variant, mode = (int(x) for x in os.environ["FUZZ_TUNING"].split(",")); tn = {512: 1, 256: 2, 128: 3}[variant] | ((mode + 1) << 2)
            seed = int(rng.integers(1, 2**31 - 1)); n = int(rng.choice([8, 20, 64, 150, 400, 1000, 2000, 3000]))
            mi = int(rng.choice([500, 3000, 20000]))
            if rng.random() < 0.6:
                ir = float(rng.uniform(0.1, 0.8)); sg = float(rng.choice([0.05, 0.1, 0.5, 1.0])); pf = float(rng.choice([0.0, 0.0, 0.6, 0.9]))
                et = int(rng.choice([0, 1])); sym = bool(rng.random() < 0.7); dg = bool(rng.random() < 0.7); th = float(rng.choice([0.5, 1.0, 2.0]))
                # [N, 6] input with laf_consistensy_coef > 0 in four cases of ten (exp_ranF.c:1394-1411, :1536-1556, :1664-1682), the
                # final LAF filter (MI_DEGENSAC_FLAG_FINAL_LAF_FILTER, :1724-1739) in half of those
                laf = bool(rng.random() < 0.4); lc = float(rng.choice([1.0, 2.0, 3.0])) if laf else 0.0
                lbad = float(rng.choice([0.1, 0.25, 0.5])); fin = int(laf and rng.random() < 0.5)
                lsig = float(rng.choice([0.05, 0.5]))
                if only is not None and case not in only: continue
                p1, p2, _, _ = syn.two_view_fundamental(n, ir, sg, seed=case, plane_fraction=pf, laf=laf, laf_bad=lbad, laf_sigma=lsig)
                tr = _Trace() if os.environ.get("FUZZ_TRACE") else None
                Mg, mg = pd.findFundamentalMatrix_(p1, p2, th, 0.9999, mi, et, sym, lc, dg, seed=seed, flags=fin, tuning=tn); sg_ = pd.last_stats()
                if tr: tr.gpu_done()
                Mo, mo, so = port.find_fundamental(p1, p2, th, 0.9999, mi, et, sym, lc, dg, seed=seed, final_laf_filter=bool(fin))
                if tr: tr.report()
                tag = f"F case={case} n={n} ir={ir:.2f} sig={sg} pf={pf} et={et} sym={sym} dg={dg} th={th} mi={mi} laf={lc} bad={lbad} lsig={lsig} fin={fin} laf_rej={so['rejected']} gpu_laf_rej={sg_['rejected']}"
                if sg_["rejected"] != so["rejected"]: sg_ = dict(sg_, samples=-1)      # the LAF check's rejections are part of the trajectory
                LAF_STATS[0] += laf; LAF_STATS[1] += so["rejected"] > 0
            else:
                if n < 8: n = 8
                ir = float(rng.uniform(0.15, 0.8)); sg = float(rng.choice([0.2, 0.5, 1.0])); laf = bool(rng.random() < 0.5)
                et = int(rng.integers(0, 5)); sym = bool(rng.random() < 0.7); th = float(rng.choice([1.0, 2.0, 4.0])); lc = 3.0 if laf else 0.0
                if only is not None and case not in only: continue
                p1, p2, _, _ = syn.homography_pairs(n, ir, sg, seed=case, laf=laf)
                Mg, mg = pd.findHomography_(p1, p2, th, 0.999, mi, et, sym, lc, seed=seed, tuning=tn); sg_ = pd.last_stats()
                Mo, mo, so = port.find_homography(p1, p2, th, 0.999, mi, et, sym, lc, seed=seed)
                tag = f"H n={n} ir={ir:.2f} sig={sg} laf={laf} et={et} sym={sym} th={th} mi={mi}"
            Mg = np.asarray(Mg, dtype=float).ravel(); Mo = np.asarray(Mo, dtype=float).ravel()
            rel = np.linalg.norm(Mg - Mo) / max(np.linalg.norm(Mo), 1e-300) if np.abs(Mo).sum() else float(np.abs(Mg).sum())
            nomodel = np.abs(Mo).sum() == 0                      # the reference leaves the mask undefined then
            res_ok = (nomodel and np.abs(Mg).sum() == 0) or (np.array_equal(np.asarray(mg, dtype=bool), np.asarray(mo, dtype=bool)) and rel < 1e-9)
            # every counter the two sides share: a path that diverges without changing the result (a skipped check, a pass more or less) shows here
            keys = ["samples", "lo_runs", "rejected", "I", "models", "best_sample"] + (["degen", "Ih", "full_passes", "ex_passes"] if "degen" in so else [])
            traj_ok = all(sg_[k_] == so[k_] for k_ in keys if k_ in so and k_ in sg_)
            if not traj_ok and verbose: print("   counters", {k_: (sg_[k_], so[k_]) for k_ in keys if k_ in so and k_ in sg_ and sg_[k_] != so[k_]})
            if not res_ok: bad_res += 1
            if not (res_ok and traj_ok):
                bad += 1
                if verbose:
                    print("MISMATCH" if not res_ok else "trajectory-only", tag, "seed", seed, "variant", variant, "mode",
                          mode, "gpu", sg_["samples"], sg_["lo_runs"], sg_["I"], "oracle", so["samples"], so["lo_runs"], so["I"], "rel", rel)
    return bad_res, bad - bad_res


def run_set_aside(N, rng_seed):
    """Random fundamental-matrix batches on a capped resident grid with long pairs set aside at random thresholds
    (tuning bits 16-31) against the same batch without it: models, masks and counters must be identical."""
    rng = np.random.default_rng(rng_seed); bad = 0; aside = 0
    for case in range(N):
        P = int(rng.integers(6, 40)); A = []; B = []
        for i in range(P):
            n = int(rng.choice([20, 150, 400, 1000, 2000])); pf = float(rng.choice([0.0, 0.0, 0.6, 0.9]))
            p1, p2, _, _ = syn.two_view_fundamental(n, float(rng.uniform(0.15, 0.8)), float(rng.choice([0.1, 0.5])), seed=1000 * case + i, plane_fraction=pf)
            A.append(p1); B.append(p2)
        seeds = [int(x) for x in rng.integers(1, 2**31 - 1, P)]; mi = int(rng.choice([3000, 20000]))
        variant = int(rng.choice([1, 2, 3])); mode = int(rng.choice([1, 2, 3])); et = int(rng.choice([0, 1])); sym = bool(rng.random() < 0.7)
        base = variant | (mode << 2) | (255 << 8)
        ets = ['sampson', 'symm_epipolar'][et]
        F0, m0 = pd.findFundamentalMatrixBatch(A, B, 0.5, 0.9999, mi, error_type=ets, symmetric_error_check=sym, seeds=seeds, tuning=base | (255 << 16)); s0 = pd.last_stats()
        tn = base | (int(rng.integers(1, 12)) << 16) | (int(rng.integers(1, 6)) << 24)
        F1, m1 = pd.findFundamentalMatrixBatch(A, B, 0.5, 0.9999, mi, error_type=ets, symmetric_error_check=sym, seeds=seeds, tuning=tn); s1 = pd.last_stats()
        aside += sum(x["set_aside"] for x in s1)
        key = lambda st: [(x["samples"], x["lo_runs"], x["models"], x["degen"], x["I"], x["best_sample"]) for x in st]
        ok = np.array_equal(np.asarray(F0), np.asarray(F1)) and all(np.array_equal(np.asarray(a), np.asarray(b)) for a, b in zip(m0, m1)) and key(s0) == key(s1)
        if not ok:
            bad += 1; print("MISMATCH set-aside case", case, "P", P, "tuning", hex(tn))
    print(f"set-aside: {N - bad}/{N} batches identical, {aside} pairs were set aside")
    return bad


def run_batches(N, rng_seed):
    """Random BATCHES through the batch entry points against the oracle, pair by pair: 2 ... 200 ragged pairs (8 ... 3000 correspondences,
    [N, 2] or [N, 6] with the LAF check), F or H, a random kernel variant / placement / resident-grid cap / set-aside threshold, and the
    per-call scheduling flags (producers on / off / with their test bits, homography helpers on / off).  Every result and every shared
    counter of every pair must be the oracle's; no pair may carry the discarded / re-run bits."""
    from pydegensac_amd import api, _lib
    rng = np.random.default_rng(rng_seed); t0 = time.time()
    bad_res = bad_traj = pairs = 0; seen = {"set_aside": 0, "streamed": 0, "F": 0, "H": 0}
    for case in range(N):
        isF = rng.random() < 0.6
        P = int(rng.choice([2, 9, 30, 80, 200], p=[0.2, 0.3, 0.25, 0.15, 0.1])); mi = int(rng.choice([500, 3000, 20000], p=[0.3, 0.4, 0.3]))
        laf = bool(rng.random() < 0.35); sym = bool(rng.random() < 0.7)
        variant = int(rng.choice([0, 1, 2, 3])); place = int(rng.choice([0, 1, 2, 3])); tn = variant | (place << 2)
        ns = [int(x) for x in rng.choice([8, 20, 64, 150, 400, 1000, 2000, 3000], P, p=[0.1, 0.1, 0.15, 0.2, 0.2, 0.15, 0.07, 0.03])]
        seeds = [int(x) for x in rng.integers(1, 2**31 - 1, P)]; A = []; B = []
        if isF:
            et = int(rng.choice([0, 1])); dg = bool(rng.random() < 0.7); th = float(rng.choice([0.5, 1.0, 2.0])); lc = float(rng.choice([1.0, 2.0, 3.0])) if laf else 0.0
            fin = int(laf and rng.random() < 0.5)
            cap = int(rng.choice([0, 0, 3, 8, 31])); sa = int(rng.choice([0, 0, 255, int(rng.integers(1, 12))]))
            tn |= (cap << 24) | (sa << 16) | ((int(rng.integers(0, 6)) << 29) if 0 < sa < 255 else 0)
            fl = int(rng.choice([0, _lib.FLAG_NO_STREAM, _lib.FLAG_STREAM_ON, _lib.FLAG_STREAM_ON | _lib.FLAG_STREAM_TEST(int(rng.integers(1, 4)))])) | fin
            for i in range(P):
                p1, p2, _, _ = syn.two_view_fundamental(ns[i], float(rng.uniform(0.1, 0.8)), float(rng.choice([0.05, 0.1, 0.5, 1.0])), seed=100000 + 1000 * case + i,
                                                        plane_fraction=float(rng.choice([0.0, 0.0, 0.6, 0.9])), laf=laf, laf_bad=float(rng.choice([0.1, 0.25, 0.5])), laf_sigma=float(rng.choice([0.05, 0.5])))
                A.append(p1); B.append(p2)
            M, masks = api._batch("F", A, B, th, 0.9999, mi, et, sym, lc, dg, seeds, 0, tn, fl); st = pd.last_stats()
            ref = [port.find_fundamental(A[i], B[i], th, 0.9999, mi, et, sym, lc, dg, seed=seeds[i], final_laf_filter=bool(fin)) for i in range(P)]
            keys = ["samples", "lo_runs", "rejected", "I", "models", "best_sample", "degen", "Ih", "full_passes", "ex_passes"]
        else:
            et = int(rng.integers(0, 5)); th = float(rng.choice([1.0, 2.0, 4.0])); lc = 3.0 if laf else 0.0
            fl = int(rng.choice([0, _lib.FLAG_NO_HJOB]))
            for i in range(P):
                p1, p2, _, _ = syn.homography_pairs(max(ns[i], 12), float(rng.uniform(0.15, 0.8)), float(rng.choice([0.2, 0.5, 1.0])), seed=100000 + 1000 * case + i, laf=laf)
                A.append(p1); B.append(p2)
            M, masks = api._batch("H", A, B, th, 0.999, mi, et, sym, lc, True, seeds, 0, tn, fl); st = pd.last_stats()
            ref = [port.find_homography(A[i], B[i], th, 0.999, mi, et, sym, lc, seed=seeds[i]) for i in range(P)]
            keys = ["samples", "lo_runs", "rejected", "I", "models", "best_sample"]
        seen["F" if isF else "H"] += 1
        for i in range(P):
            Mo, mo, so = ref[i]; Mg = np.asarray(M[i], float).ravel(); Mo = np.asarray(Mo, float).ravel(); pairs += 1
            seen["set_aside"] += st[i].get("set_aside", 0); seen["streamed"] += st[i].get("streamed", 0)
            nomodel = np.abs(Mo).sum() == 0
            rel = np.linalg.norm(Mg - Mo) / max(np.linalg.norm(Mo), 1e-300) if not nomodel else float(np.abs(Mg).sum())
            res_ok = (nomodel and np.abs(Mg).sum() == 0) or (np.array_equal(masks[i], np.asarray(mo, bool)) and rel < 1e-9)
            diff = {k_: (st[i][k_], so[k_]) for k_ in keys if k_ in so and k_ in st[i] and st[i][k_] != so[k_]}
            if st[i].get("discarded") or st[i].get("rerun"): diff["discarded/rerun"] = (st[i].get("discarded"), st[i].get("rerun"))
            if not res_ok or diff:
                bad_res += not res_ok; bad_traj += bool(res_ok)
                print("MISMATCH" if not res_ok else "trajectory-only", "batch", case, "F" if isF else "H", "pair", i, "of", P, "n", ns[i], "mi", mi, "et", et, "sym", sym,
                      "laf", lc, "th", th, "tuning", hex(tn), "flags", hex(fl), "seed", seeds[i], "rel", rel, diff)
    print(f"batches: {N} ({seen['F']} F, {seen['H']} H), {pairs} pairs: results differ in {bad_res}, trajectory counters only in {bad_traj}; "
          f"{seen['set_aside']} pairs were set aside, {seen['streamed']} fed by a producer; {time.time() - t0:.0f} s")
    return bad_res, bad_traj


def _compare(tag, Mg, mg, sg_, Mo, mo, so, keys):
    Mg = np.asarray(Mg, dtype=float).ravel(); Mo = np.asarray(Mo, dtype=float).ravel()
    nomodel = np.abs(Mo).sum() == 0
    rel = np.linalg.norm(Mg - Mo) / max(np.linalg.norm(Mo), 1e-300) if not nomodel else float(np.abs(Mg).sum())
    res_ok = (nomodel and np.abs(Mg).sum() == 0) or (np.array_equal(np.asarray(mg, dtype=bool), np.asarray(mo, dtype=bool)) and rel < 1e-9)
    diff = {k_: (sg_[k_], so[k_]) for k_ in keys if k_ in so and k_ in sg_ and sg_[k_] != so[k_]}
    if not res_ok or diff: print("MISMATCH" if not res_ok else "trajectory-only", tag, "rel", rel, diff, flush=True)
    return res_ok, not diff


def run_edges(N, rng_seed):
    """The corners the main sweep does not visit: 8 ... 200 correspondences, sample budgets of 1 ... 600 (around the 50-sample rule of the
    first local optimisation, exp_ranF.c:1497-1500, and the 256-sample chunk), confidences 0.5 ... 0.999999, thresholds 0.05 ... 100 px,
    noise-free and pixel-quantised coordinates (exact ties and zeros), repeated correspondences (rank-deficient samples), all-inlier sets."""
    rng = np.random.default_rng(rng_seed); t0 = time.time(); bad_res = bad_traj = 0
    only = os.environ.get("FUZZ_ONLY"); only = set(int(x) for x in only.split(",")) if only else None
    for case in range(N):
        isF = rng.random() < 0.6; variant = int(rng.choice([512, 256, 128])); mode = int(rng.choice([0, 1, 2]))
        tn = {512: 1, 256: 2, 128: 3}[variant] | ((mode + 1) << 2)
        n = int(rng.choice([8, 9, 10, 11, 12, 16, 30, 64, 200])); mi = int(rng.choice([0, 1, 2, 7, 49, 50, 51, 52, 100, 255, 256, 257, 600]))
        conf = float(rng.choice([0.0, 0.5, 0.9, 0.99, 0.9999, 0.999999, 1.0])); th = float(rng.choice([0.0, 0.05, 0.5, 2.0, 10.0, 100.0]))
        ir = float(rng.choice([0.2, 0.5, 0.8, 1.0])); sg = float(rng.choice([0.0, 0.1, 1.0])); seed = int(rng.integers(1, 2**31 - 1))
        quant = bool(rng.random() < 0.3); dup = bool(rng.random() < 0.25); laf = bool(rng.random() < 0.3); sym = bool(rng.random() < 0.7)
        et = int(rng.integers(0, 2 if isF else 5)); dg = bool(rng.random() < 0.7); pf = float(rng.choice([0.0, 0.6, 1.0]))
        kdup = int(rng.integers(2, max(3, n // 2))); lafc = float(rng.choice([1.0, 3.0]))
        if only is not None and case not in only: continue
        if isF: p1, p2, _, _ = syn.two_view_fundamental(n, ir, sg, seed=case, plane_fraction=pf, laf=laf)
        else:
            n = max(n, 11)                                   # <= 10: the reference's 4-point u2h path reads uninitialised memory
            p1, p2, _, _ = syn.homography_pairs(n, ir, sg, seed=case, laf=laf)
        if quant: p1[:, :2] = np.round(p1[:, :2]); p2[:, :2] = np.round(p2[:, :2])
        if dup: p1[1:kdup] = p1[0]; p2[1:kdup] = p2[0]
        lc = lafc if laf else 0.0
        tag = f"edge case={case} {'F' if isF else 'H'} n={n} mi={mi} conf={conf} th={th} ir={ir} sig={sg} quant={quant} dup={dup} laf={lc} sym={sym} et={et} dg={dg} pf={pf} seed={seed} variant={variant} mode={mode}"
        if isF:
            Mg, mg = pd.findFundamentalMatrix_(p1, p2, th, conf, mi, et, sym, lc, dg, seed=seed, tuning=tn); sg_ = pd.last_stats()
            Mo, mo, so = port.find_fundamental(p1, p2, th, conf, mi, et, sym, lc, dg, seed=seed)
            keys = ["samples", "lo_runs", "rejected", "I", "models", "best_sample", "degen", "Ih", "full_passes", "ex_passes"]
        else:
            Mg, mg = pd.findHomography_(p1, p2, th, conf, mi, et, sym, lc, seed=seed, tuning=tn); sg_ = pd.last_stats()
            Mo, mo, so = port.find_homography(p1, p2, th, conf, mi, et, sym, lc, seed=seed)
            keys = ["samples", "lo_runs", "rejected", "I", "models", "best_sample"]
        r, t = _compare(tag, Mg, mg, sg_, Mo, mo, so, keys); bad_res += not r; bad_traj += (r and not t)
    print(f"edges: {N} cases: results differ in {bad_res}, trajectory counters only in {bad_traj}; {time.time() - t0:.0f} s")
    return bad_res, bad_traj


def run_legacy(N, rng_seed):
    """The reference's older fundamental-matrix drivers (exp_ransacF / exp_ransacFcustom, MI_DEGENSAC_FLAG_LEGACY_F): main-sweep sizes and
    corner budgets, with and without exp_ransacFcustom's own symmetric check."""
    from pydegensac_amd import _lib
    rng = np.random.default_rng(rng_seed); t0 = time.time(); bad_res = bad_traj = 0
    for case in range(N):
        variant = int(rng.choice([512, 256, 128])); mode = int(rng.choice([0, 1, 2])); tn = {512: 1, 256: 2, 128: 3}[variant] | ((mode + 1) << 2)
        n = int(rng.choice([8, 12, 30, 64, 150, 400, 1000, 2000, 3000])); mi = int(rng.choice([1, 7, 49, 50, 51, 257, 500, 3000, 20000]))
        ir = float(rng.uniform(0.1, 0.9)); sg = float(rng.choice([0.05, 0.1, 0.5, 1.0])); pf = float(rng.choice([0.0, 0.0, 0.6, 0.9])); seed = int(rng.integers(1, 2**31 - 1))
        et = int(rng.choice([0, 1])); sym = bool(rng.random() < 0.5); th = float(rng.choice([0.5, 1.0, 2.0])); conf = float(rng.choice([0.9, 0.9999]))
        p1, p2, _, _ = syn.two_view_fundamental(n, ir, sg, seed=20000 + case, plane_fraction=pf)
        Mg, mg = pd.findFundamentalMatrix_(p1, p2, th, conf, mi, et, sym, 0.0, True, seed=seed, flags=_lib.FLAG_LEGACY_F, tuning=tn); sg_ = pd.last_stats()
        Mo, mo, so = port.find_fundamental(p1, p2, th, conf, mi, et, sym, 0.0, True, seed=seed, legacy=True)
        tag = f"legacy case={case} n={n} mi={mi} ir={ir:.3f} sig={sg} pf={pf} et={et} sym={sym} th={th} conf={conf} seed={seed} variant={variant} mode={mode}"
        r, t = _compare(tag, Mg, mg, sg_, Mo, mo, so, ["samples", "lo_runs", "I", "models", "best_sample", "degen", "Ih", "full_passes", "ex_passes"]); bad_res += not r; bad_traj += (r and not t)
    print(f"legacy: {N} cases: results differ in {bad_res}, trajectory counters only in {bad_traj}; {time.time() - t0:.0f} s")
    return bad_res, bad_traj


def run_large(N, rng_seed):
    """5 000 ... 50 000 correspondences, up to 100 000 samples: the cooperative multi-workgroup mode (n >= 8192), long inlier lists in the
    local optimisations (random subsets beyond inlLimit), the hash table at its capacity."""
    rng = np.random.default_rng(rng_seed); t0 = time.time(); bad_res = bad_traj = 0
    for case in range(N):
        isF = rng.random() < 0.65; n = int(rng.choice([5000, 9000, 20000, 50000], p=[0.3, 0.3, 0.25, 0.15])); mi = int(rng.choice([3000, 20000, 100000], p=[0.4, 0.4, 0.2]))
        ir = float(rng.uniform(0.08, 0.7)); sg = float(rng.choice([0.1, 0.5, 1.0])); seed = int(rng.integers(1, 2**31 - 1)); sym = bool(rng.random() < 0.7)
        laf = bool(rng.random() < 0.3); et = int(rng.integers(0, 2 if isF else 5)); th = float(rng.choice([0.5, 1.0, 2.0])); lc = 3.0 if laf else 0.0
        tag = f"large case={case} {'F' if isF else 'H'} n={n} mi={mi} ir={ir:.3f} sig={sg} laf={lc} sym={sym} et={et} th={th} seed={seed}"
        if isF:
            pf = float(rng.choice([0.0, 0.0, 0.6])); dg = bool(rng.random() < 0.7)
            p1, p2, _, _ = syn.two_view_fundamental(n, ir, sg, seed=5000 + case, plane_fraction=pf, laf=laf)
            Mg, mg = pd.findFundamentalMatrix_(p1, p2, th, 0.9999, mi, et, sym, lc, dg, seed=seed); sg_ = pd.last_stats()
            Mo, mo, so = port.find_fundamental(p1, p2, th, 0.9999, mi, et, sym, lc, dg, seed=seed)
            keys = ["samples", "lo_runs", "rejected", "I", "models", "best_sample", "degen", "Ih", "full_passes", "ex_passes"]; tag += f" pf={pf} dg={dg}"
        else:
            p1, p2, _, _ = syn.homography_pairs(n, ir, sg, seed=5000 + case, laf=laf)
            Mg, mg = pd.findHomography_(p1, p2, th, 0.999, mi, et, sym, lc, seed=seed); sg_ = pd.last_stats()
            Mo, mo, so = port.find_homography(p1, p2, th, 0.999, mi, et, sym, lc, seed=seed)
            keys = ["samples", "lo_runs", "rejected", "I", "models", "best_sample"]
        r, t = _compare(tag + f" threads={sg_.get('threads')} placement={sg_.get('placement')}", Mg, mg, sg_, Mo, mo, so, keys); bad_res += not r; bad_traj += (r and not t)
    print(f"large: {N} cases: results differ in {bad_res}, trajectory counters only in {bad_traj}; {time.time() - t0:.0f} s")
    return bad_res, bad_traj


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] in ("edges", "large", "legacy"):
        br, bt = {"edges": run_edges, "large": run_large, "legacy": run_legacy}[sys.argv[1]](int(sys.argv[2]) if len(sys.argv) > 2 else 40, int(sys.argv[3]) if len(sys.argv) > 3 else 0)
        sys.exit(1 if br else 0)
    if len(sys.argv) > 1 and sys.argv[1] == "batches":
        br, bt = run_batches(int(sys.argv[2]) if len(sys.argv) > 2 else 40, int(sys.argv[3]) if len(sys.argv) > 3 else 0)
        sys.exit(1 if br else 0)
    if len(sys.argv) > 1 and sys.argv[1] == "set-aside":
        sys.exit(1 if run_set_aside(int(sys.argv[2]) if len(sys.argv) > 2 else 40, int(sys.argv[3]) if len(sys.argv) > 3 else 0) else 0)
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    t0 = time.time()
    br, bt = run(N, int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    print(f"{N - br - bt}/{N} identical (results differ in {br}, trajectory counters only in {bt}) in {time.time() - t0:.1f} s; "
          f"F + LAF cases {LAF_STATS[0]}, with candidates rejected by the LAF check {LAF_STATS[1]}")
