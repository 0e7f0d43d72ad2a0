"""PCIe-inclusive rate of the host-pointer batch API (numpy in, numpy out: pinned staging, one H2D, the kernel, one D2H)
on the bench workload: C2 x P pairs.  DESIGN.md quotes it next to the HBM-resident `value` of bench.py."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import pydegensac_amd as pd
from pydegensac_amd import synthetic as syn, parallel
P = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
A = []; B = []
for i in range(P):
    p1, p2, _, _ = syn.two_view_fundamental(2000, 0.4, 0.1, seed=i); A.append(p1); B.append(p2)
seeds = parallel.pair_seeds(0, P)
best = 1e9
for it in range(3):
    t = time.perf_counter()
    F, m = pd.findFundamentalMatrixBatch(A, B, 0.5, 0.9999, 100000, seeds=seeds)
    dt = time.perf_counter() - t; best = min(best, dt)
models = sum(s["models"] for s in pd.last_stats())
print(f"host batch API: {P} pairs, best of 3 = {best * 1e3:.1f} ms wall (incl. numpy concatenation + PCIe), {models / best / 1e6:.1f} M models/s")
