"""C3 (BASELINE config 3): findHomography, 5000 correspondences with LAFs, LAF + symmetric checks, LO on.
Batch of P pairs through the batch API (host staging); per-pair kernel time from the stats block; one pair
checked against the oracle and timed on the CPU reference when oracle/_ref is present."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import pydegensac_amd as pd
from pydegensac_amd import synthetic as syn
P = int(sys.argv[1]) if len(sys.argv) > 1 else 64
n = int(sys.argv[2]) if len(sys.argv) > 2 else 5000
A = []; B = []
for i in range(P):
    p1, p2, lab, _ = syn.homography_pairs(n, 0.4, 0.5, seed=i, laf=True); A.append(p1); B.append(p2)
for rep in range(2):
    t = time.perf_counter()
    out = pd.findHomographyBatch(A, B, 1.0, 0.999, 50000, 3.0, "sampson", True, seeds=list(range(1, P + 1)))
    dt = time.perf_counter() - t
st = pd.last_stats()
tk = np.array([s_["ticks_total"] for s_ in st]) / 1e5; sm = np.array([s_["samples"] for s_ in st]); md = np.array([s_["models"] for s_ in st])
print("batch wall %.1f ms for %d pairs; per-pair kernel ms mean %.2f max %.2f; samples mean %.0f; models mean %.0f" % (dt * 1e3, P, tk.mean(), tk.max(), sm.mean(), md.mean()))
from oracle import port
Hg, mg = pd.findHomography_(A[0], B[0], 1.0, 0.999, 50000, 0, True, 3.0, seed=1)
st0 = pd.last_stats(); print('gpu pair 0: kernel %.2f ms' % (st0['ticks_total'] / 1e5), {k: st0[k] for k in ['samples', 'lo_runs', 'models', 'I']})
t = time.perf_counter(); Ho, mo, so = port.find_homography(A[0], B[0], 1.0, 0.999, 50000, 0, True, 3.0, seed=1); dto = time.perf_counter() - t
print("oracle pair 0: %.1f ms" % (dto * 1e3), {k: so[k] for k in ["samples", "lo_runs", "models", "I"]})
try:
    print("mask diff", int((np.asarray(mg) != mo).sum()), "relH", float(np.linalg.norm(Hg / np.linalg.norm(Hg) - Ho / np.linalg.norm(Ho))))
except Exception as e:
    print("compare failed:", e)
