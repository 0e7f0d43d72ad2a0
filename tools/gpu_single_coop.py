"""One C2 pair per call: the automatic choice (512 threads, pair in LDS, a producer from the start) against the cooperative mode forced
onto the same pair (placement HBM, k helper workgroups: the chunk's scoring and the local optimisation's repetitions as claimable units)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import pydegensac_amd as pd
from pydegensac_amd import synthetic as syn, _lib
ks = [int(x) for x in sys.argv[1:]] or [0, 7, 15, 23]
pairs = [syn.two_view_fundamental(2000, 0.4, 0.1, seed=s)[:2] for s in range(12)]
ref = None
for k in ks:
    tune = 0 if k == 0 else (_lib.TUNE_LATENCY | _lib.TUNE_PLACE_HBM | _lib.TUNE_HELPERS(k))
    ts = []; res = []
    for rep in range(3):
        for i, (p1, p2) in enumerate(pairs):
            t = time.perf_counter(); F, m = pd.findFundamentalMatrix_(p1, p2, 0.5, 0.9999, 100000, 0, True, 0.0, True, seed=i + 1, tuning=tune) if False else pd.findFundamentalMatrixBatch([p1], [p2], seeds=[i + 1], tuning=tune)
            ts.append((time.perf_counter() - t) * 1e3)
            if rep == 0: res.append((np.asarray(F[0]).copy(), np.asarray(m[0]).copy()))
    same = "" if ref is None else " identical: %s" % all(np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) for a, b in zip(ref, res))
    ref = ref or res
    ts = np.array(ts[len(pairs):])
    print(f"helpers {k:2d}: median {np.median(ts):6.2f} ms  mean {ts.mean():6.2f}  min {ts.min():5.2f}  max {ts.max():6.2f}{same}", flush=True)
