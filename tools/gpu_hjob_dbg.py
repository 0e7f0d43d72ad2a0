import sys, os; sys.path.insert(0, '/root/repo')
import numpy as np
import pydegensac_amd as pd
from pydegensac_amd import _lib, synthetic as syn
from oracle import port
A, B = [], []
for i, (n, ir) in enumerate([(5000, 0.4), (1200, 0.3), (3000, 0.5), (400, 0.6), (2500, 0.2), (800, 0.4)]):
    p1, p2, _, _ = syn.homography_pairs(n, ir, 0.5, seed=230 + i, laf=True); A.append(p1); B.append(p2)
seeds = [41 + i for i in range(len(A))]
ora = [port.find_homography(A[p], B[p], 2.0, 0.999, 50000, 0, True, 3.0, seed=seeds[p]) for p in range(len(A))]
MODES = [int(x) for x in sys.argv[1:]] or [1, 0]
for variant, name in ((2, '256'), (1, '512'), (3, '128')):
    for mode in MODES:
        _lib.set_hjob_mode(mode); bad = 0; badres = 0
        for rep in range(8):
            try:
                H, m = pd.findHomographyBatch(A, B, 2.0, 0.999, 50000, 3.0, "sampson", True, seeds=seeds, tuning=variant | (1 << 2))
            except Exception as e:
                print("   call failed:", e); bad += 100; continue
            st = pd.last_stats()
            for p in range(len(A)):
                so = ora[p][2]
                if (st[p]["samples"], st[p]["lo_runs"], st[p]["models"]) != (so["samples"], so["lo_runs"], so["models"]):
                    bad += 1; print("  ", name, mode, rep, "pair", p, "models", st[p]["models"], "oracle", so["models"], "I", st[p]["I"], so["I"])
                if not np.array_equal(np.asarray(m[p]), ora[p][1]): badres += 1
        print("variant", name, "helpers", mode, "counter mismatches", bad, "mask mismatches", badres, flush=True)
