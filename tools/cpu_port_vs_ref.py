"""CPU check: oracle port vs unmodified reference (oracle/_ref) on seeded synthetic pairs: stats, mask, model."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import port, ref
from pydegensac_amd import synthetic as syn
bad = 0; tot = 0
for dseed in range(int(sys.argv[1]) if len(sys.argv) > 1 else 12):
    for n, ir in ((2000, 0.4), (500, 0.3)):
        p1, p2, _, _ = syn.two_view_fundamental(n, ir, 0.1, seed=dseed)
        for seed in (1, 7, 42):
            Fp, mp, sp = port.find_fundamental(p1, p2, seed=seed)
            Fr, mr, sr = ref.find_fundamental(p1, p2, seed=seed)
            tot += 1
            same = np.array_equal(mp, mr) and sp["samples"] == sr["samples"] and sp["lo_runs"] == sr["lo_runs"]
            rel = np.linalg.norm(Fp - Fr) / max(np.linalg.norm(Fr), 1e-300)
            if not same or rel > 1e-9: bad += 1; print("MISMATCH", dseed, n, seed, sp, sr, rel)
print(f"{tot - bad}/{tot} identical")
