"""A/B timing of the homography kernel inside ONE process: C3 x 1024 (and x 256) with helper workgroups on / off, one C3 pair per call."""
import sys, os, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pydegensac_amd import synthetic as syn, _lib, parallel
import pydegensac_amd as pd
L = _lib.lib(); N = 5000; dev = torch.device('cuda', 0)
def data(P):
    a = np.empty((P * N, 6)); b = np.empty((P * N, 6))
    for i in range(P):
        p1, p2 = syn.homography_pairs(N, 0.4, 0.5, seed=i, laf=True)[:2]; a[i*N:(i+1)*N] = p1; b[i*N:(i+1)*N] = p2
    offs = np.arange(P + 1, dtype=np.int64) * N
    return (torch.from_numpy(a).to(dev), torch.from_numpy(b).to(dev), torch.from_numpy(offs).to(dev), offs,
            torch.from_numpy(parallel.pair_seeds(0, P).astype(np.int64)).to(dev).to(torch.int32))
TUNES = [int(x, 0) for x in sys.argv[1:]] or [0]
def run(P, d, reps=3, tune=0):
    d_a, d_b, d_off, offs, d_seeds = d
    d_H = torch.zeros((P, 9), dtype=torch.float64, device=dev); d_mask = torch.zeros(P * N, dtype=torch.uint8, device=dev); d_st = torch.zeros((P, 16), dtype=torch.int32, device=dev)
    prm = _lib.make_params(2.0, 0.999, 50000, 0, True, 3.0, True, 0, tune)
    ts = []
    for it in range(reps + 1):
        torch.cuda.synchronize(); t = time.perf_counter()
        L.mi_degensac_find_homography_batch_dev(d_a.data_ptr(), d_b.data_ptr(), d_off.data_ptr(), offs.ctypes.data_as(C.POINTER(C.c_int64)), P, 6, C.byref(prm),
                                                d_seeds.data_ptr(), 0, None, d_H.data_ptr(), d_mask.data_ptr(), d_st.data_ptr())
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t) * 1e3)
    st = d_st.cpu().numpy()
    return min(ts[1:]), float(np.mean(ts[1:])), int(st[0, 14]), int(st[0, 15]), float(st[:, 13].max() / 1e5), float(st[:, 13].mean() / 1e5), d_H.cpu().numpy(), d_mask.cpu().numpy()
PS = [int(x) for x in os.environ.get('AB_H_PAIRS', '1024,256').split(',')]
for P in PS:
    d = data(P); ref = None
    for tune in TUNES:
      for mode in ((1, 0) if 'AB_H_PAIRS' not in os.environ else (1,)):
        _lib.set_hjob_mode(mode)
        best, mean, thr, plc, longest, meanp, H, m = run(P, d, tune=tune)
        same = "" if ref is None else " identical: %s" % (np.array_equal(ref[0], H) and np.array_equal(ref[1], m))
        ref = ref or (H, m)
        print(f"C3 x {P:4d} helpers {mode}: best {best:6.2f} ms mean {mean:6.2f} ms  threads {thr} placement {plc}  longest pair {longest:5.1f} ms mean pair {meanp:5.2f} ms{same}", flush=True)
p1, p2 = syn.homography_pairs(N, 0.4, 0.5, seed=0, laf=True)[:2]
for mode in ((1, 0) if 'AB_H_PAIRS' not in os.environ else ()):
    _lib.set_hjob_mode(mode); ts = []
    for r in range(12):
        t = time.perf_counter(); pd.findHomography_(p1, p2, 2.0, 0.999, 50000, 0, True, 3.0, seed=r + 1); ts.append((time.perf_counter() - t) * 1e3)
    print(f"one C3 pair per call, helpers {mode}: median {np.median(ts[1:]):5.2f} ms  min {min(ts[1:]):5.2f}  max {max(ts[1:]):5.2f}", flush=True)
