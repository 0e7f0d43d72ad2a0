"""Homography-only GPU-vs-oracle sweep aimed at the local optimisation (one repetition per wave): large point
sets, low inlier ratios (several LO runs sharing one hash table), every metric, LAF on / off, every workgroup size and
placement, and the serial order (tuning bit 5) now and then.  Masks, sample / LO / scored-model counters identical, models to 1e-9.
    python tools/gpu_fuzz_h.py [n_cases] [seed]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import pydegensac_amd as pd
from pydegensac_amd import synthetic as syn
from oracle import port
N = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
bad = 0; t0 = time.time(); lo_hist = {}
for case in range(N):
    variant = int(rng.choice([512, 256, 128])); mode = int(rng.choice([0, 1, 2])); serial = 32 if rng.random() < 0.15 else 0
    tn = {512: 1, 256: 2, 128: 3}[variant] | ((mode + 1) << 2) | serial
    seed = int(rng.integers(1, 2**31 - 1)); n = int(rng.choice([16, 40, 300, 1200, 3000, 5000, 8000]))
    ir = float(rng.uniform(0.08, 0.6)); sg = float(rng.choice([0.3, 0.5, 1.0, 2.0])); laf = bool(rng.random() < 0.5)
    et = int(rng.integers(0, 5)); sym = bool(rng.random() < 0.7); th = float(rng.choice([1.0, 2.0, 4.0])); lc = 3.0 if laf else 0.0
    mi = int(rng.choice([2000, 20000, 50000]))
    p1, p2, _, _ = syn.homography_pairs(n, ir, sg, seed=7000 + case, laf=laf)
    Mg, mg = pd.findHomography_(p1, p2, th, 0.999, mi, et, sym, lc, seed=seed, tuning=tn); sg_ = pd.last_stats()
    Mo, mo, so = port.find_homography(p1, p2, th, 0.999, mi, et, sym, lc, seed=seed)
    Mg = np.asarray(Mg, float).ravel(); Mo = np.asarray(Mo, float).ravel()
    rel = np.linalg.norm(Mg - Mo) / max(np.linalg.norm(Mo), 1e-300) if np.abs(Mo).sum() else float(np.abs(Mg).sum())
    nomodel = np.abs(Mo).sum() == 0
    ok = ((nomodel and np.abs(Mg).sum() == 0) or (np.array_equal(np.asarray(mg, bool), np.asarray(mo, bool)) and rel < 1e-9)) and \
         (sg_["samples"], sg_["lo_runs"], sg_["models"], sg_["rejected"]) == (so["samples"], so["lo_runs"], so["models"], so["rejected"])
    lo_hist[so["lo_runs"]] = lo_hist.get(so["lo_runs"], 0) + 1
    if not ok:
        bad += 1
        print("MISMATCH n=%d ir=%.2f sig=%s laf=%s et=%d sym=%s th=%s mi=%d seed=%d variant=%d mode=%d serial=%d" % (n, ir, sg, laf, et, sym, th, mi, seed, variant, mode, serial),
              "gpu", sg_["samples"], sg_["lo_runs"], sg_["models"], sg_["I"], "oracle", so["samples"], so["lo_runs"], so["models"], so["I"], "rel", rel)
print("%d/%d identical in %.1f s; LO runs per case: %s" % (N - bad, N, time.time() - t0, dict(sorted(lo_hist.items()))))
