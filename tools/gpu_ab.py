"""A/B timing inside ONE process on ONE box (boxes differ by +-5 %): C2 batches of 512 and 4096 pairs, every configuration in
turn, three launches each (best and mean), plus single calls.  usage: gpu_ab.py [configs...]  with configs out of
  base            stream mode on (default), repetitions one per wave
  nostream        stream mode off
  serial          stream off, local optimisation / innerH repetitions in the serial order (round 3's kernel structure)
"""
import sys, os, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pydegensac_amd import synthetic as syn, _lib, parallel
import pydegensac_amd as pd
L = _lib.lib()
cfgs = sys.argv[1:] or ["base", "nostream", "serial"]
N = 2000
dev = torch.device('cuda', 0)
def data(P):
    a = np.empty((P * N, 2)); b = np.empty((P * N, 2))
    for i in range(P):
        p1, p2, _, _ = syn.two_view_fundamental(N, 0.4, 0.1, seed=i); a[i*N:(i+1)*N] = p1; b[i*N:(i+1)*N] = p2
    offs = np.arange(P + 1, dtype=np.int64) * N
    return (torch.from_numpy(a).to(dev), torch.from_numpy(b).to(dev), torch.from_numpy(offs).to(dev), offs,
            torch.from_numpy(parallel.pair_seeds(0, P).astype(np.int64)).to(dev).to(torch.int32))
def run(P, d, tuning, reps=3):
    d_a, d_b, d_off, offs, d_seeds = d
    d_F = torch.zeros((P, 9), dtype=torch.float64, device=dev); d_mask = torch.zeros(P * N, dtype=torch.uint8, device=dev); d_st = torch.zeros((P, 16), dtype=torch.int32, device=dev)
    prm = _lib.make_params(0.5, 0.9999, 100000, 0, True, 0.0, True, 0, tuning)
    ts = []
    for it in range(reps + 1):
        torch.cuda.synchronize(); t = time.perf_counter()
        rc = L.mi_degensac_find_fundamental_batch_dev(d_a.data_ptr(), d_b.data_ptr(), d_off.data_ptr(), offs.ctypes.data_as(C.POINTER(C.c_int64)), P, 2, C.byref(prm),
                                                      d_seeds.data_ptr(), 0, None, d_F.data_ptr(), d_mask.data_ptr(), d_st.data_ptr())
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t) * 1e3)
    st = d_st.cpu().numpy()
    return min(ts[1:]), float(np.mean(ts[1:])), int(((st[:, 15] >> 9) & 1).sum()), float(st[:, 13].max() / 1e5), d_F.cpu().numpy(), d_mask.cpu().numpy()
ref = {}
for P in (512, 4096):
    d = data(P)
    for cfg in cfgs:
        _lib.set_stream_mode(-1 if cfg == "base" else 0)
        tn = _lib.TUNE_F_SERIAL_REPS if cfg == "serial" else 0
        best, mean, streamed, longest, F, m = run(P, d, tn)
        same = ""
        if P in ref: same = " results identical to the first configuration: %s" % (np.array_equal(ref[P][0], F) and np.array_equal(ref[P][1], m))
        else: ref[P] = (F, m)
        print(f"P={P:5d} {cfg:9s} best {best:7.2f} ms  mean {mean:7.2f} ms  streamed pairs {streamed:4d}  longest pair {longest:6.1f} ms{same}", flush=True)
p1, p2 = syn.two_view_fundamental(N, 0.4, 0.1, seed=0)[:2]
for cfg in cfgs:
    _lib.set_stream_mode(-1 if cfg == "base" else 0)
    tn = _lib.TUNE_F_SERIAL_REPS if cfg == "serial" else 0
    ts = []
    for r in range(12):
        t = time.perf_counter(); pd.findFundamentalMatrix_(p1, p2, 0.5, 0.9999, 100000, 0, True, 0.0, True, seed=r + 1, tuning=tn); ts.append((time.perf_counter() - t) * 1e3)
    print(f"single call {cfg:9s} median {np.median(ts[1:]):6.2f} ms  min {min(ts[1:]):6.2f}  max {max(ts[1:]):6.2f}", flush=True)
