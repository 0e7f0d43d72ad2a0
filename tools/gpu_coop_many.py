"""Cooperative mode with MORE pairs than owner slots (every owner runs several pairs one after the other, its helpers follow): a batch
of 2000-point pairs with 23 / 7 helpers per owner at 512 / 256 / 128 threads, every pair against the CPU oracle."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import pydegensac_amd as pd
from pydegensac_amd import synthetic as syn, _lib
from oracle import port
P = int(sys.argv[1]) if len(sys.argv) > 1 else 48
A, B = [], []
for i in range(P):
    p1, p2, _, _ = syn.two_view_fundamental(600 + 37 * (i % 40), 0.35 + 0.01 * (i % 20), 0.1, seed=200 + i, plane_fraction=0.6 if i % 5 == 0 else 0.0); A.append(p1); B.append(p2)
seeds = list(range(11, 11 + P))
ora = [port.find_fundamental(A[p], B[p], 0.5, 0.9999, 20000, seed=seeds[p]) for p in range(P)]
bad = 0
for variant, k in ((_lib.TUNE_LATENCY, 23), (_lib.TUNE_LATENCY, 7), (_lib.TUNE_THROUGHPUT, 23), (_lib.TUNE_THROUGHPUT4, 7)):
    F, m = pd.findFundamentalMatrixBatch(A, B, max_iters=20000, seeds=seeds, tuning=variant | _lib.TUNE_PLACE_HBM | _lib.TUNE_HELPERS(k))
    st = pd.last_stats()
    n_bad = 0
    for p, (Fo, mo, so) in enumerate(ora):
        ok = (st[p]["samples"], st[p]["lo_runs"], st[p]["models"]) == (so["samples"], so["lo_runs"], so["models"]) and np.array_equal(np.asarray(m[p]), mo.astype(bool)) \
             and np.linalg.norm(np.asarray(F[p]).ravel() - np.asarray(Fo).ravel()) <= 1e-9 * max(np.linalg.norm(Fo), 1e-300)
        n_bad += 0 if ok else 1
    print(f"variant {variant} helpers {k}: {P - n_bad}/{P} pairs identical to the oracle", flush=True); bad += n_bad
print("bad", bad)
