"""ransacH2el GPU-vs-oracle sweep: random ellipse-correspondence sets (9 … 6000 correspondences, 0-70 % inliers, noise on points and
frames), LO on / off, inlLimit 0 / 16 / 40, ragged batches.  Masks, sample / LO / scored-model counters identical, models to 1e-9.
    python tools/gpu_fuzz_h2el.py [n_batches] [seed]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import pydegensac_amd as pd
from pydegensac_amd import synthetic as syn
from oracle import port
N = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
bad = 0; tot = 0; t0 = time.time()
for b in range(N):
    P = int(rng.integers(1, 12)); U = []
    for i in range(P):
        n = int(rng.choice([9, 14, 30, 100, 400, 1500, 3000, 6000])); ir = float(rng.choice([0.0, 0.1, 0.2, 0.4, 0.7]))
        U.append(syn.ellipse_pairs(n, ir, float(rng.choice([0.3, 1.0, 2.0])), 100000 + 100 * b + i, float(rng.choice([0.0, 0.05, 0.2])))[0])
    seeds = [int(x) for x in rng.integers(1, 2**31 - 1, P)]
    do_lo = bool(rng.random() < 0.8); lim = int(rng.choice([0, 0, 16, 40])); th = float(rng.choice([0.01, 1.0, 4.0, 9.0], p=[0.1, 0.3, 0.3, 0.3])); mi = int(rng.choice([1, 3, 49, 50, 51, 200, 2000, 10000], p=[0.05, 0.05, 0.04, 0.04, 0.04, 0.26, 0.26, 0.26]))
    conf = float(rng.choice([0.95, 0.99, 0.999]))
    H, m = pd.ransacH2el_batch(U, th, conf, mi, do_lo, lim, seeds=seeds, raw=True); st = pd.last_stats()
    for p in range(P):
        if lim and U[p].shape[0] <= 14:
            continue                                              # 4-point u2h of the reference reads uninitialised memory (Htools.c:108-114)
        Ho, mo, so = port.ransacH2el(U[p], th, conf, mi, do_lo, lim, seeds[p]); tot += 1
        a = np.asarray(H[p]).ravel(); o = np.asarray(Ho).ravel()
        rel = np.linalg.norm(a - o) / max(np.linalg.norm(o), 1e-300) if np.abs(o).sum() else float(np.abs(a).sum())
        nomodel = np.abs(o).sum() == 0                            # nothing found: the reference's mask is whatever its uninitialised errs[3] holds
        ok = ((st[p]["samples"], st[p]["lo_runs"], st[p]["I"], st[p]["models"]) == (so["samples"], so["lo_runs"], so["I"], so["models"])
              and ((nomodel and np.abs(a).sum() == 0) or (np.array_equal(np.asarray(m[p]), mo) and rel < 1e-9)))
        if not ok:
            bad += 1; print("MISMATCH batch", b, "pair", p, "n", U[p].shape[0], "lo", do_lo, "lim", lim, "th", th, "mi", mi, "seed", seeds[p], st[p]["samples"], st[p]["lo_runs"], st[p]["I"], so, rel)
print("%d/%d pairs identical in %.1f s" % (tot - bad, tot, time.time() - t0))
