"""Scheduling probe: a batch of P C2 pairs (distinct data) at several set-aside settings.
usage: gpu_sched.py P out_dir tuning_word...   (tuning words as in include/mi_degensac.h; 0 = automatic)
Prints best-of-3 batch time per setting, checks that results do not depend on it, and saves the per-pair stats blocks."""
import sys, os, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pydegensac_amd import synthetic as syn, _lib, parallel
L = _lib.lib()
P = int(sys.argv[1]); out = sys.argv[2]; tunes = [int(x, 0) for x in sys.argv[3:]] or [0]
N = 2000
a = np.empty((P * N, 2)); b = np.empty((P * N, 2))
for i in range(P):
    p1, p2, _, _ = syn.two_view_fundamental(N, 0.4, 0.1, seed=i); a[i*N:(i+1)*N] = p1; b[i*N:(i+1)*N] = p2
offs = np.arange(P + 1, dtype=np.int64) * N
dev = torch.device('cuda', 0)
d_a = torch.from_numpy(a).to(dev); d_b = torch.from_numpy(b).to(dev); d_off = torch.from_numpy(offs).to(dev)
d_seeds = torch.from_numpy(parallel.pair_seeds(0, P).astype(np.int64)).to(dev).to(torch.int32)
d_F = torch.zeros((P, 9), dtype=torch.float64, device=dev); d_mask = torch.zeros(P * N, dtype=torch.uint8, device=dev); d_st = torch.zeros((P, 16), dtype=torch.int32, device=dev)
ref = None
os.makedirs(out, exist_ok=True)
for tn in tunes:
    prm = _lib.make_params(0.5, 0.9999, 100000, 0, True, 0.0, True, 0, tn)
    best = 1e9
    for it in range(3):
        torch.cuda.synchronize(); t = time.perf_counter()
        rc = L.mi_degensac_find_fundamental_batch_dev(d_a.data_ptr(), d_b.data_ptr(), d_off.data_ptr(), offs.ctypes.data_as(C.POINTER(C.c_int64)), P, 2, C.byref(prm),
                                                      d_seeds.data_ptr(), 0, None, d_F.data_ptr(), d_mask.data_ptr(), d_st.data_ptr())
        torch.cuda.synchronize(); best = min(best, time.perf_counter() - t)
    st = d_st.cpu().numpy(); F = d_F.cpu().numpy(); m = d_mask.cpu().numpy()
    key = (F.tobytes(), m.tobytes(), st[:, :12].tobytes())
    same = "ref" if ref is None else ("same" if key == ref else "DIFFERENT")
    if ref is None: ref = key
    tk = st[:, 13] / 1e5
    aside = (st[:, 15] >> 8) & 1
    print(f"tuning {tn:#x}: rc {rc} batch {best*1e3:.1f} ms  models/s {st[:,4].sum()/best/1e6:.2f}M  threads {st[0,14]} set aside {int(aside.sum())}  pair ms mean {tk.mean():.2f} max {tk.max():.1f}  sum/512 {tk.sum()/512:.1f}  results {same}", flush=True)
    np.save(os.path.join(out, f"sched_st_{tn:x}.npy"), st)
