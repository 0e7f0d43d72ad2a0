"""Feasibility probe for mixed-width scheduling: do a 256-thread launch (1 workgroup per CU) and a 128-thread launch (2 per CU) of the
fundamental-matrix kernel run SIDE BY SIDE on one device (two streams), and what do their pairs cost then?
usage: gpu_mix.py [pairs] [share of the pairs that goes to the wide launch]"""
import os as _os
_os.environ.setdefault("MI_DEGENSAC_LIB", _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "libmi_degensac_dev.so"))
import sys, os, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pydegensac_amd import synthetic as syn, _lib, parallel
L = _lib.lib()
P = int(sys.argv[1]) if len(sys.argv) > 1 else 4096; share = float(sys.argv[2]) if len(sys.argv) > 2 else 1 / 3
N = 2000; dev = torch.device('cuda', 0)
a = np.empty((P * N, 2)); b = np.empty((P * N, 2))
for i in range(P):
    p1, p2, _, _ = syn.two_view_fundamental(N, 0.4, 0.1, seed=i); a[i*N:(i+1)*N] = p1; b[i*N:(i+1)*N] = p2
d_a = torch.from_numpy(a).to(dev); d_b = torch.from_numpy(b).to(dev)
seeds = torch.from_numpy(parallel.pair_seeds(0, P).astype(np.int64)).to(dev).to(torch.int32)
Pw = int(P * share)
class Part:
    def __init__(s, lo, hi, tuning, percu):
        s.lo, s.hi, s.n = lo, hi, hi - lo; s.percu = percu
        s.offs = np.arange(s.n + 1, dtype=np.int64) * N; s.d_off = torch.from_numpy(s.offs).to(dev)
        s.F = torch.zeros((s.n, 9), dtype=torch.float64, device=dev); s.m = torch.zeros(s.n * N, dtype=torch.uint8, device=dev); s.st = torch.zeros((s.n, 16), dtype=torch.int32, device=dev)
        s.prm = _lib.make_params(0.5, 0.9999, 100000, 0, True, 0.0, True, 0, tuning); s.stream = torch.cuda.Stream(dev)
        s.e0 = torch.cuda.Event(enable_timing=True); s.e1 = torch.cuda.Event(enable_timing=True)
    def launch(s):
        L.mi_degensac_dev_set_per_cu(s.percu)
        s.e0.record(s.stream)
        _lib.check(L.mi_degensac_find_fundamental_batch_dev(d_a.data_ptr() + s.lo * N * 16, d_b.data_ptr() + s.lo * N * 16, s.d_off.data_ptr(), s.offs.ctypes.data_as(C.POINTER(C.c_int64)), s.n, 2,
                   C.byref(s.prm), seeds.data_ptr() + 4 * s.lo, 0, C.c_void_p(s.stream.cuda_stream), s.F.data_ptr(), s.m.data_ptr(), s.st.data_ptr()))
        s.e1.record(s.stream)
    def report(s, tag):
        st = s.st.cpu().numpy(); busy = st[:, 13].astype(np.float64) / 1e5
        print(f"  {tag:28s} pairs {s.n:5d}  kernel {s.e0.elapsed_time(s.e1):7.2f} ms  mean pair {busy.mean():6.2f} ms  longest {busy.max():6.1f} ms  threads {int(st[0, 14])}")
wide = Part(0, Pw, 2, 1); narrow = Part(Pw, P, 3, 2)
for rnd in range(2):
    for name, parts in (("wide alone", [wide]), ("narrow alone", [narrow]), ("both", [wide, narrow])):
        torch.cuda.synchronize(); t = time.perf_counter()
        for p_ in parts: p_.launch()
        torch.cuda.synchronize(); dt = (time.perf_counter() - t) * 1e3
        print(f"{name}: wall {dt:.2f} ms")
        for p_ in parts: p_.report("wide (256 thr, 1 per CU)" if p_ is wide else "narrow (128 thr, 2 per CU)")
L.mi_degensac_dev_set_per_cu(0)
full = Part(0, P, 0, 0)
for rnd in range(2):
    torch.cuda.synchronize(); t = time.perf_counter(); full.launch(); torch.cuda.synchronize(); print(f"today's launch: wall {(time.perf_counter() - t) * 1e3:.2f} ms"); full.report("256 thr, 2 per CU")
