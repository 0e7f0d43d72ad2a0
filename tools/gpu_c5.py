"""C5 stress (BASELINE config 5): 50000 correspondences, 10 % inliers, max_iters 200000, conf 0.9999 — the
points do not fit LDS, so this runs the global-memory variant of the kernel.  Checks against the oracle."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import pydegensac_amd as pd
from pydegensac_amd import synthetic as syn
from oracle import port
n = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
mi = int(sys.argv[2]) if len(sys.argv) > 2 else 200000
p1, p2, lab, _ = syn.two_view_fundamental(n, 0.1, 0.1, seed=0)
t = time.perf_counter(); F, m = pd.findFundamentalMatrix(p1, p2, 0.5, 0.9999, mi, seed=1); dt = time.perf_counter() - t
st = pd.last_stats()
print("GPU", {k: st[k] for k in ["samples", "lo_runs", "degen", "models", "I"]}, f"{dt:.2f} s wall, kernel {st['ticks_total'] / 1e8:.2f} s, models/s {st['models'] / (st['ticks_total'] / 1e8):.0f}")
if os.environ.get("C5_ORACLE", "1") == "1":
    t = time.perf_counter(); Fo, mo, so = port.find_fundamental(p1, p2, 0.5, 0.9999, mi, seed=1); dto = time.perf_counter() - t
    print("ORA", {k: so[k] for k in ["samples", "lo_runs", "degen", "models", "I"]}, f"{dto:.2f} s")
    a = F / np.linalg.norm(F); b = Fo / np.linalg.norm(Fo)
    print("mask diff", int((np.asarray(m) != mo).sum()), "relF", float(np.linalg.norm(a - b)), "recall", float((np.asarray(m) & lab).sum() / lab.sum()))
