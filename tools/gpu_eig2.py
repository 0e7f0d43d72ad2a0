"""Two 9 x 9 eigen-problems per wave (dg_eig2.h, mi_degensac_mat3 op 5) against one per wave (op 3): bit-for-bit equality on the unit test's
matrices in random pairings, and the time per problem of both (ops 6 / 7: every wave repeats its solve from fresh copies)."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pydegensac_amd import _lib
L = _lib.lib(); dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
rng = np.random.default_rng(33)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
A = np.zeros((N, 9, 9))
for t in range(N):
    k = t % 5
    if k == 0: m = rng.normal(size=(rng.integers(8, 60), 9)); a = m.T @ m
    elif k == 1:
        x1 = rng.normal(size=(14, 2)) * np.sqrt(2) / 2; x2 = x1 + rng.normal(size=(14, 2)) * 0.05
        m = np.stack([np.r_[b[0] * np.r_[a_, 1.0], b[1] * np.r_[a_, 1.0], np.r_[a_, 1.0]] for a_, b in zip(x1, x2)]); a = m.T @ m
    elif k == 2: a = rng.normal(size=(9, 9)) * 10.0 ** rng.integers(-5, 6); a = a + a.T
    elif k == 3: m = rng.normal(size=(6, 9)); a = m.T @ m
    else: a = np.diag(rng.normal(size=9)) + 1e-9 * rng.normal(size=(9, 9)); a = (a + a.T) / 2
    A[t] = (a + a.T) / 2
def run(op, M, reps=0):
    n = len(M); out = np.zeros((n, 90)); flag = np.zeros(n, np.int32); flag[0] = reps
    _lib.check(L.mi_degensac_mat3(op, dp(np.ascontiguousarray(M)), n, 0, dp(out), flag.ctypes.data_as(C.POINTER(C.c_int32))))
    return out, flag
bad = 0
for trial in range(3):
    perm = rng.permutation(N) if trial else np.arange(N)      # pairings: neighbours of the same kind first, then random ones
    o1, f1 = run(3, A[perm]); o2, f2 = run(5, A[perm])
    same = np.array_equal(o1, o2) and np.array_equal(f1, f2)
    nb = int((o1 != o2).any(axis=1).sum())
    bad += nb
    print(f"pairing {trial}: op 5 == op 3 bit for bit: {same} (differing problems: {nb} of {N})", flush=True)
# timing: like kinds side by side (k = 1: the estimator's own matrices) and random pairings
R = 200
for name, M in (("estimator-like (14-point Gram matrices)", A[1::5][:512]), ("mixed kinds, random pairs", A[rng.permutation(N)[:512]])):
    t1, _ = run(6, M, R); t2, _ = run(7, M, R)
    n = len(M)
    us1 = t1[:n, 0].ravel()[:n] if False else t1.ravel()[:n] / 100.0 / R          # ticks of 10 ns -> us per solve
    us2 = t2.ravel()[:(n + 1) // 2] / 100.0 / R
    print(f"{name}: one problem per wave {us1.mean():6.2f} us per solve; two per wave {us2.mean():6.2f} us per PAIR = {us2.mean() / 2:6.2f} us per problem "
          f"(two solves in a row: {2 * us1.mean():6.2f} us)", flush=True)
print("RESULT", "identical" if bad == 0 else f"{bad} differ")
