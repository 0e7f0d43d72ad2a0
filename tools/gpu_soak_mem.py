"""Device-memory soak: a few hundred batch launches of alternating kinds and sizes (F / H / ransacH2el, 1 ... 600 pairs, host-pointer and
device-pointer entry points, two host threads); free device memory after the warm-up launches must not keep falling (the library caches
one workspace per stream and shrinks it after 8 oversized launches).
    python tools/gpu_soak_mem.py [rounds]"""
import os, sys, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import pydegensac_amd as pd
from pydegensac_amd import synthetic as syn

R = int(sys.argv[1]) if len(sys.argv) > 1 else 300
rng = np.random.default_rng(0)
F = [syn.two_view_fundamental(int(n), 0.4, 0.3, seed=i)[:2] for i, n in enumerate(rng.choice([100, 500, 2000], 64))]
H = [syn.homography_pairs(int(n), 0.4, 0.5, seed=i)[:2] for i, n in enumerate(rng.choice([100, 1000, 5000], 64))]
E = [syn.ellipse_pairs(int(n), 0.3, 1.0, i)[0] for i, n in enumerate(rng.choice([50, 500, 3000], 32))]
BIG = [syn.two_view_fundamental(22000 + 1000 * i, 0.3, 0.1, seed=70 + i)[:2] for i in range(3)]       # one large pair per call: cooperative + fan mode (round 6)
free = lambda: torch.cuda.mem_get_info(0)[0] / 2**20
def work(tid, rounds, log):
    r = np.random.default_rng(tid)
    for it in range(rounds):
        k = int(r.integers(0, 3)); P = int(r.choice([1, 3, 40, 600]))
        if it % 17 == 5:
            b = BIG[int(r.integers(0, len(BIG)))]; pd.findFundamentalMatrixBatch([b[0]], [b[1]], 0.5, 0.9999, 4000, seeds=[it])
        elif k == 0:
            ids = r.integers(0, len(F), P); pd.findFundamentalMatrixBatch([F[i][0] for i in ids], [F[i][1] for i in ids], 0.5, 0.9999, 3000, seeds=list(range(P)))
        elif k == 1:
            ids = r.integers(0, len(H), P); pd.findHomographyBatch([H[i][0] for i in ids], [H[i][1] for i in ids], 1.0, 0.999, 2000, seeds=list(range(P)))
        else:
            ids = r.integers(0, len(E), min(P, 64)); pd.ransacH2el_batch([E[i] for i in ids], 4.0, 0.99, 1000, True, 0, seeds=list(range(len(ids))))
        if tid == 0 and it % 25 == 0: log.append((it, free()))
f0 = free(); log = []; ends = []
for phase in range(3):                                   # the SAME sequence of launches three times: the caches end each phase in the same state
    ts = [threading.Thread(target=work, args=(t, R, log)) for t in range(2)]
    [t.start() for t in ts]; [t.join() for t in ts]
    torch.cuda.synchronize(); ends.append(free())
print("free MiB before", round(f0), "while running", sorted(set(round(v) for _, v in log)), "after each of three identical phases", [round(v) for v in ends])
# workspaces grow with a 600-pair launch and shrink again after eight small ones, so free memory moves by 1-2 GiB WITHIN a phase; a leak shows
# as less free memory at the end of a later phase
print("OK: nothing leaks" if ends[2] >= ends[0] - 64 and ends[1] >= ends[0] - 64 else "LEAK?", "2 x", R, "launches per phase")
