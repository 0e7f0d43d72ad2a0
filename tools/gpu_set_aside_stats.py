"""Setting long pairs aside (tuning bits 16-23): how many pairs of a C2 batch were set aside at a given threshold, what the
others cost, and how long the set-aside ones waited.  usage: gpu_set_aside_stats.py [pairs] [threshold in units of 256 samples]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import pydegensac_amd as pd
from pydegensac_amd import synthetic as syn, parallel
P = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
A = []; B = []
for i in range(P):
    p1, p2, _, _ = syn.two_view_fundamental(2000, 0.4, 0.1, seed=i); A.append(p1); B.append(p2)
for thr in [int(x) for x in sys.argv[2:]] or [255, 32]:
    for rep in range(2):
        F, m = pd.findFundamentalMatrixBatch(A, B, 0.5, 0.9999, 100000, seeds=parallel.pair_seeds(0, P), tuning=thr << 16)
    st = pd.last_stats()
    s = np.array([x["samples"] for x in st]); t = np.array([x["ticks_total"] for x in st]) / 1e5; a = np.array([x["set_aside"] for x in st]).astype(bool)
    heavy = s >= 100000
    print(f"threshold {thr}: set aside {int(a.sum())} (heavy among them {int((a & heavy).sum())}), heavy not set aside {int((heavy & ~a).sum())}; "
          f"run time of pairs not set aside: sum/512 = {t[~a].sum() / 512:.1f} ms, max {t[~a].max():.1f} ms; "
          f"set-aside pairs start-to-end: mean {t[a].mean() if a.any() else 0:.1f} min {t[a].min() if a.any() else 0:.1f} max {t[a].max() if a.any() else 0:.1f} ms")
    if not a.any():
        print(f"   heavy: {int(heavy.sum())} pairs, mean {t[heavy].mean():.1f} ms; light mean {t[~heavy].mean():.2f} ms, p99 {np.percentile(t[~heavy], 99):.1f}, samples p50 {np.percentile(s[~heavy], 50):.0f} p99 {np.percentile(s[~heavy], 99):.0f} max {s[~heavy].max()}")
