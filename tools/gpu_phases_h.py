"""Per-pair phase times of the homography kernel on the C3 workload (development build's 100 MHz timers):
solve / score||sample / commit / LO, and inside the LO: passes, long-list least squares + eigen-solve, inlier-set hash,
small fits, consistency checks.  usage: gpu_phases_h.py [pairs]"""
import os as _os
_os.environ.setdefault("MI_DEGENSAC_LIB", _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "libmi_degensac_dev.so"))
import sys, numpy as np, ctypes as C, torch, time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pydegensac_amd import synthetic as syn, _lib, parallel
L = _lib.lib()
P = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
N = 5000
a = np.empty((P * N, 6)); b = np.empty((P * N, 6))
for i in range(P):
    p1, p2, _, _ = syn.homography_pairs(N, 0.4, 0.5, seed=i, laf=True); a[i*N:(i+1)*N] = p1; b[i*N:(i+1)*N] = p2
offs = np.arange(P + 1, dtype=np.int64) * N
dev = torch.device('cuda', 0)
d_a = torch.from_numpy(a).to(dev); d_b = torch.from_numpy(b).to(dev); d_off = torch.from_numpy(offs).to(dev)
d_seeds = torch.from_numpy(parallel.pair_seeds(0, P).astype(np.int64)).to(dev).to(torch.int32)
d_F = torch.zeros((P, 9), dtype=torch.float64, device=dev); d_mask = torch.zeros(P * N, dtype=torch.uint8, device=dev); d_st = torch.zeros((P, 16), dtype=torch.int32, device=dev)
d_ph = torch.zeros((P, 16), dtype=torch.int64, device=dev)
L.mi_degensac_debug_phases(C.c_void_p(d_ph.data_ptr()))
prm = _lib.make_params(2.0, 0.999, 50000, 0, True, 3.0, True)
for it in range(2):
    torch.cuda.synchronize(); t = time.perf_counter()
    rc = L.mi_degensac_find_homography_batch_dev(d_a.data_ptr(), d_b.data_ptr(), d_off.data_ptr(), offs.ctypes.data_as(C.POINTER(C.c_int64)), P, 6, C.byref(prm), d_seeds.data_ptr(), 0, None, d_F.data_ptr(), d_mask.data_ptr(), d_st.data_ptr())
    torch.cuda.synchronize(); dt = time.perf_counter() - t
print("rc", rc, "batch ms", dt * 1e3)
ph = d_ph.cpu().numpy().astype(np.float64) / 1e5; st = d_st.cpu().numpy()
names = ["solve", "score||sample", "commit", "LO", "-", "-", "tail", "total", "LO passes", "LO lsq+eig (long lists)", "LO hash", "LO small fits", "LO checks|gather", "lsq_par", "eig", "(mark)"]
print("mean ms per pair:", {n: round(float(ph[:, i].mean()), 3) for i, n in enumerate(names) if n not in ("-", "(mark)")})
print("samples mean", st[:, 0].mean(), "lo_runs mean", st[:, 1].mean(), "threads", st[0, 14], "placement", st[0, 15] & 255)
# the longest pairs of the launch (a one-generation launch ends with its slowest pair)
order = np.argsort(-ph[:, 7])[:8]
for p in order:
    print("pair", int(p), {n: round(float(ph[p, i]), 2) for i, n in enumerate(names) if n not in ("-", "(mark)", "lsq_par", "eig")}, "samples", int(st[p, 0]), "lo_runs", int(st[p, 1]), "models", int(st[p, 4]), "best_sample", int(st[p, 7]))
