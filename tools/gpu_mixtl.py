"""(Needs the library of commit d779705: the mixed-width launch was measured slower and removed again, profiles/r6_ab_mixed_width.log.)
Timeline of a mixed-width launch (development build): when do the pairs of each kernel end, when do the long pairs start?
usage: gpu_mixtl.py [pairs] [wide grid or per-CU] [narrow grid or per-CU] [rule]"""
import os as _os
_os.environ.setdefault("MI_DEGENSAC_LIB", _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "libmi_degensac_dev.so"))
import sys, os, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pydegensac_amd import synthetic as syn, _lib, parallel
L = _lib.lib()
P = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
mw = int(sys.argv[2]) if len(sys.argv) > 2 else 1; mn = int(sys.argv[3]) if len(sys.argv) > 3 else 2; rule = int(sys.argv[4], 0) if len(sys.argv) > 4 else 0
flags = int(sys.argv[5], 0) if len(sys.argv) > 5 else 0
L.mi_degensac_dev_set_mix(-1, -rule, mw, mn)
N = 2000; G = 4096
a = np.empty((P * N, 2)); b = np.empty((P * N, 2))
for i in range(P):
    p1, p2, _, _ = syn.two_view_fundamental(N, 0.4, 0.1, seed=i); a[i*N:(i+1)*N] = p1; b[i*N:(i+1)*N] = p2
offs = np.arange(P + 1, dtype=np.int64) * N
dev = torch.device('cuda', 0)
d_a = torch.from_numpy(a).to(dev); d_b = torch.from_numpy(b).to(dev); d_off = torch.from_numpy(offs).to(dev)
d_seeds = torch.from_numpy(parallel.pair_seeds(0, P).astype(np.int64)).to(dev).to(torch.int32)
d_F = torch.zeros((P, 9), dtype=torch.float64, device=dev); d_mask = torch.zeros(P * N, dtype=torch.uint8, device=dev); d_st = torch.zeros((P, 16), dtype=torch.int32, device=dev)
d_ph = torch.zeros((2 * P + G, 16), dtype=torch.int64, device=dev)
L.mi_degensac_debug_phases(C.c_void_p(d_ph.data_ptr()))
prm = _lib.make_params(0.5, 0.9999, 100000, 0, True, 0.0, True, flags, 0)
for it in range(2):
    d_ph.zero_(); torch.cuda.synchronize(); t = time.perf_counter()
    _lib.check(L.mi_degensac_find_fundamental_batch_dev(d_a.data_ptr(), d_b.data_ptr(), d_off.data_ptr(), offs.ctypes.data_as(C.POINTER(C.c_int64)), P, 2, C.byref(prm),
                                             d_seeds.data_ptr(), 0, None, d_F.data_ptr(), d_mask.data_ptr(), d_st.data_ptr()))
    torch.cuda.synchronize(); dt = time.perf_counter() - t
ph = d_ph.cpu().numpy(); st = d_st.cpu().numpy()
wg = ph[P:P + G]; m = wg[:, 0] != 0
t_exit = wg[m, 0].astype(np.float64) / 1e5; t_start = wg[m, 10].astype(np.float64) / 1e5; role = wg[m, 11]
t0 = t_start[t_start > 0].min()
print(f"wall {dt * 1e3:.1f} ms; workgroups {m.sum()}")
for r_, nm in ((0, "single"), (1, "wide"), (2, "narrow")):
    k = role == r_
    if k.sum(): print(f"  {nm:6s} workgroups {int(k.sum()):4d}: start min {t_start[k].min() - t0:6.2f} p50 {np.median(t_start[k]) - t0:6.2f} max {t_start[k].max() - t0:6.2f} ms;  exit min {t_exit[k].min() - t0:6.1f} p50 {np.median(t_exit[k]) - t0:6.1f} max {t_exit[k].max() - t0:6.1f} ms")
pend = ph[:P, 15].astype(np.float64) / 1e5 - t0; busy = st[:, 13] / 1e5; thr = st[:, 14]; longp = st[:, 0] >= 99999
for t_ in sorted(set(thr.tolist())):
    k = thr == t_
    print(f"  pairs on {t_} threads: {int(k.sum())}; last end {pend[k].max():6.1f} ms; long pairs {int((k & longp).sum())}")
k = longp
if k.sum():
    s_ = pend[k] - busy[k]
    print(f"  long pairs: {int(k.sum())}; busy mean {busy[k].mean():.1f} max {busy[k].max():.1f}; (end - busy) = latest possible start: p10 {np.percentile(s_, 10):.1f} p50 {np.median(s_):.1f} p90 {np.percentile(s_, 90):.1f} max {s_.max():.1f}; end p50 {np.median(pend[k]):.1f} p90 {np.percentile(pend[k], 90):.1f} max {pend[k].max():.1f}")
order = np.argsort(-pend)[:12]
for i in order: print(f"    pair {i:5d} end {pend[i]:6.1f} busy {busy[i]:5.1f} samples {st[i, 0]:6d} lo {st[i, 1]:2d} degen {st[i, 5]} threads {thr[i]} aside {(st[i, 15] >> 8) & 1}")
