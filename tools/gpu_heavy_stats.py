"""Which C2 pairs are the slow ones?  Batch of P bench pairs: samples, DEGENSAC events, LO runs and kernel time per pair."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import pydegensac_amd as pd
from pydegensac_amd import synthetic as syn, parallel
P = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
A = []; B = []
for i in range(P):
    p1, p2, _, _ = syn.two_view_fundamental(2000, 0.4, 0.1, seed=i); A.append(p1); B.append(p2)
F, m = pd.findFundamentalMatrixBatch(A, B, 0.5, 0.9999, 100000, seeds=parallel.pair_seeds(0, P))
st = pd.last_stats()
s = np.array([x["samples"] for x in st]); d = np.array([x["degen"] for x in st]); t = np.array([x["ticks_total"] for x in st]) / 1e5
lo = np.array([x["lo_runs"] for x in st]); ih = np.array([x["Ih"] for x in st])
heavy = s >= 100000
print("pairs", P, "heavy", int(heavy.sum()), "heavy with degen==0", int((heavy & (d == 0)).sum()), "light with degen>0", int((~heavy & (d > 0)).sum()),
      "light with Ih>0 (checksample positive at least once)", int((~heavy & (ih > 0)).sum()), "heavy with Ih>0", int((heavy & (ih > 0)).sum()))
print("light: mean ms %.1f p95 %.1f max %.1f ; heavy: mean ms %.1f max %.1f" % (t[~heavy].mean(), np.percentile(t[~heavy], 95), t[~heavy].max(), t[heavy].mean(), t[heavy].max()))
print("light samples: mean %.0f p95 %.0f max %d ; lo_runs light mean %.1f heavy mean %.1f" % (s[~heavy].mean(), np.percentile(s[~heavy], 95), s[~heavy].max(), lo[~heavy].mean(), lo[heavy].mean()))
mid = ~heavy & (s > 8192)
print("light pairs with more than 8192 samples:", int(mid.sum()), "their mean ms %.1f" % (t[mid].mean() if mid.any() else 0))
