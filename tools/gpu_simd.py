"""Which SIMDs do the waves of the co-resident workgroups of the REAL fundamental-matrix kernel sit on?  (development build: every
workgroup records HW_ID / XCC_ID of its waves when it leaves.)   usage: gpu_simd.py [pairs] [tuning] [per_cu]"""
import os as _os
_os.environ.setdefault("MI_DEGENSAC_LIB", _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "libmi_degensac_dev.so"))
import sys, os, collections, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pydegensac_amd import synthetic as syn, _lib, parallel
L = _lib.lib()
P = int(sys.argv[1]) if len(sys.argv) > 1 else 2048; tn = int(sys.argv[2], 0) if len(sys.argv) > 2 else 0
if len(sys.argv) > 3: L.mi_degensac_dev_set_per_cu(int(sys.argv[3]))
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
_lib.check(L.mi_degensac_find_fundamental_batch_dev(d_a.data_ptr(), d_b.data_ptr(), d_off.data_ptr(), offs.ctypes.data_as(C.POINTER(C.c_int64)), P, 2, C.byref(prm),
                                                    d_seeds.data_ptr(), 0, None, d_F.data_ptr(), d_mask.data_ptr(), d_st.data_ptr()))
torch.cuda.synchronize()
ph = d_ph.cpu().numpy(); st = d_st.cpu().numpy()
nw = int(st[0, 14]) // 64
rows = ph[P:P + G]; rows = rows[rows[:, 0] != 0]
hw = rows[:, 1:1 + nw] & 0xffffffff; xcc = (rows[:, 1:1 + nw] >> 32) & 0xf
simd = (hw >> 4) & 3; cu = (hw >> 8) & 15; sh = (hw >> 12) & 1; se = (hw >> 13) & 7
per = collections.defaultdict(list)
for b_ in range(len(rows)): per[(int(xcc[b_, 0]), int(se[b_, 0]), int(sh[b_, 0]), int(cu[b_, 0]))].append(tuple(int(x) for x in simd[b_]))
print("threads", nw * 64, "workgroups", len(rows), "CUs used", len(per), "workgroups per CU", sorted(collections.Counter(len(v) for v in per.values()).items()))
w0 = collections.Counter(tuple(sorted(collections.Counter(w[0] for w in v).values(), reverse=True)) for v in per.values())
print("wave-0 (the serial wave) SIMD sharing per CU: multiplicities of the SIMDs that hold a wave 0 -> number of CUs")
for k, c in w0.most_common(8): print("  ", k, c)
pat = collections.Counter(tuple(sorted(v)) for v in per.values())
print("most common per-CU patterns (SIMD of every wave, per workgroup):")
for k, c in pat.most_common(6): print("  ", c, k)
