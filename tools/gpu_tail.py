"""The tail of a launch (development build): when does each resident workgroup run out of work?
usage: gpu_tail.py [pairs] [tuning]   -> kernel time, mean / percentiles of the workgroups' exit times, the idle share"""
import os as _os
_os.environ.setdefault("MI_DEGENSAC_LIB", _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "libmi_degensac_dev.so"))
import sys, os, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pydegensac_amd import synthetic as syn, _lib, parallel
L = _lib.lib()
P = int(sys.argv[1]) if len(sys.argv) > 1 else 4096; tn = int(sys.argv[2], 0) if len(sys.argv) > 2 else 0
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
prm = _lib.make_params(0.5, 0.9999, 100000, 0, True, 0.0, True, 0, tn)
for it in range(2):
    d_ph.zero_(); torch.cuda.synchronize(); t = time.perf_counter()
    L.mi_degensac_find_fundamental_batch_dev(d_a.data_ptr(), d_b.data_ptr(), d_off.data_ptr(), offs.ctypes.data_as(C.POINTER(C.c_int64)), P, 2, C.byref(prm),
                                             d_seeds.data_ptr(), 0, None, d_F.data_ptr(), d_mask.data_ptr(), d_st.data_ptr())
    torch.cuda.synchronize(); dt = time.perf_counter() - t
ph = d_ph.cpu().numpy(); st = d_st.cpu().numpy()
ex = ph[P:P + G, 0]; ex = ex[ex != 0].astype(np.float64) / 1e5
end = ex.max(); rel = end - ex
busy = st[:, 13].astype(np.float64).sum() / 1e5
print(f"pairs {P} wall {dt*1e3:.1f} ms; workgroups {len(ex)}; sum of pair busy time / workgroups = {busy/len(ex):.1f} ms")
print("workgroup idle before the end of the launch, ms: mean %.1f  p50 %.1f  p90 %.1f  max %.1f  -> %.1f %% of the launch" % (rel.mean(), np.percentile(rel, 50), np.percentile(rel, 90), rel.max(), 100 * rel.mean() / (dt * 1e3)))
pe = ph[:P, 15].astype(np.float64) / 1e5; order = np.argsort(-pe)[:24]
print("last pairs to finish (ms before the end, busy ms, samples, lo runs, degen, set aside):")
for i in order: print("   %.1f  busy %.1f  samples %d  lo %d  degen %d  aside %d" % (end - pe[i], st[i, 13] / 1e5, st[i, 0], st[i, 1], st[i, 5], (st[i, 15] >> 8) & 1))
last = np.argsort(-st[:, 13])[:8]
print("longest pairs (busy ms, samples, lo runs, degen):", [(round(st[i, 13] / 1e5, 1), int(st[i, 0]), int(st[i, 1]), int(st[i, 5])) for i in last])

pk = ph[P + G:, :5].astype(np.float64); aside = pk[:, 3] > 0
busy_all = st[:, 13].astype(np.float64) / 1e5
rem = busy_all - pk[:, 3] / 1e5                       # busy time after the pair was set aside
q0 = aside & (pk[:, 0] < 8192)
print("set aside: %d, of them in the 'few samples left' queue: %d; remaining busy ms there: mean %.1f p50 %.1f p90 %.1f p99 %.1f max %.1f" % (aside.sum(), q0.sum(), rem[q0].mean(), *np.percentile(rem[q0], [50, 90, 99]), rem[q0].max()))
X = np.stack([pk[q0, 0], pk[q0, 1], pk[q0, 2], pk[q0, 3] / 1e5, pk[q0, 4], np.ones(q0.sum())], 1); y = rem[q0]
for name, col in (("samples left", 0), ("LO runs so far", 1), ("degen so far", 2), ("busy so far", 3), ("I so far", 4)):
    print("  corr(remaining busy, %s) = %.3f" % (name, np.corrcoef(X[:, col], y)[0, 1]))
w, *_ = np.linalg.lstsq(X, y, rcond=None); pred = X @ w
print("  linear fit on all five: corr %.3f; weights" % np.corrcoef(pred, y)[0, 1], np.round(w, 4))
os.makedirs(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "gpurun_out", "tail"), exist_ok=True)
np.save(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "gpurun_out", "tail", "park.npy"), np.concatenate([pk, busy_all[:, None], st[:, :6].astype(np.float64)], 1))
