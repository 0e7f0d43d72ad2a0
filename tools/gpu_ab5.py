"""A/B timing inside ONE process on ONE box: C2 batches, every configuration in turn, `reps` launches each (best and mean).
usage: gpu_ab5.py P[,P2...] name=flags:tuning[:lo_width[:knob[:per_cu]]] ...      e.g.  gpu_ab5.py 4096 base=0:0 stream=8:0 t128=0:3 t128x2=0:3:-1:0:2
(per_cu caps the resident workgroups per CU: development build)
Results of every configuration are compared with the first one's (must be identical)."""
import sys, os, time, ctypes as C
# the lo_width / knob fields need the development build (make -C pydegensac_amd/csrc dev); flags and tuning work on the product library
if any(a.count(":") >= 2 for a in sys.argv[2:]):
    os.environ.setdefault("MI_DEGENSAC_LIB", os.path.join(os.path.dirname(os.path.abspath(__file__)), "libmi_degensac_dev.so"))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pydegensac_amd import synthetic as syn, _lib, parallel
L = _lib.lib()
Ps = [int(x) for x in sys.argv[1].split(",")]
cfgs = []
for a in sys.argv[2:]:
    name, rest = a.split("=", 1); parts = rest.split(":")
    cfgs.append((name, int(parts[0], 0), int(parts[1], 0) if len(parts) > 1 else 0, int(parts[2]) if len(parts) > 2 else -1, int(parts[3], 0) if len(parts) > 3 else 0, int(parts[4]) if len(parts) > 4 else 0,
                 int(parts[5], 0) if len(parts) > 5 else 0, int(parts[6]) if len(parts) > 6 else 1, int(parts[7]) if len(parts) > 7 else 2))      # mixed-width: rule word, wide / narrow workgroups per CU
N = 2000
dev = torch.device('cuda', 0)
def data(P):
    a = np.empty((P * N, 2)); b = np.empty((P * N, 2))
    for i in range(P):
        p1, p2, _, _ = syn.two_view_fundamental(N, 0.4, 0.1, seed=i); a[i*N:(i+1)*N] = p1; b[i*N:(i+1)*N] = p2
    offs = np.arange(P + 1, dtype=np.int64) * N
    return (torch.from_numpy(a).to(dev), torch.from_numpy(b).to(dev), torch.from_numpy(offs).to(dev), offs,
            torch.from_numpy(parallel.pair_seeds(0, P).astype(np.int64)).to(dev).to(torch.int32))
def run(P, d, flags, tuning, reps=3):
    d_a, d_b, d_off, offs, d_seeds = d
    d_F = torch.zeros((P, 9), dtype=torch.float64, device=dev); d_mask = torch.zeros(P * N, dtype=torch.uint8, device=dev); d_st = torch.zeros((P, 16), dtype=torch.int32, device=dev)
    prm = _lib.make_params(0.5, 0.9999, 100000, 0, True, 0.0, True, flags, tuning)
    ts = []
    for it in range(reps + 1):
        torch.cuda.synchronize(); t = time.perf_counter()
        rc = L.mi_degensac_find_fundamental_batch_dev(d_a.data_ptr(), d_b.data_ptr(), d_off.data_ptr(), offs.ctypes.data_as(C.POINTER(C.c_int64)), P, 2, C.byref(prm),
                                                      d_seeds.data_ptr(), 0, None, d_F.data_ptr(), d_mask.data_ptr(), d_st.data_ptr())
        _lib.check(rc)
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t) * 1e3)
    st = d_st.cpu().numpy()
    busy = float(st[:, 13].astype(np.float64).sum() / 1e5)
    return min(ts[1:]), float(np.mean(ts[1:])), int(((st[:, 15] >> 9) & 1).sum()), float(st[:, 13].max() / 1e5), busy, d_F.cpu().numpy(), d_mask.cpu().numpy(), st
for P in Ps:
    d = data(P); ref = None
    for rnd in range(2):                                  # every configuration twice, interleaved (drift of the box shows)
        for name, flags, tuning, low, knob, percu, rule, mw, mn in cfgs:
            if hasattr(L, "mi_degensac_dev_set_mix"): L.mi_degensac_dev_set_mix(-1 if (mw != 1 or mn != 2 or rule or name.startswith("mix")) else 0, -rule, mw, mn)      # mixed-width launches are off by default
            if hasattr(L, "mi_degensac_dev_set_per_cu"): L.mi_degensac_dev_set_per_cu(percu)
            if hasattr(L, "mi_degensac_dev_set_lo_width"): L.mi_degensac_dev_set_lo_width(low)
            if hasattr(L, "mi_degensac_dev_set_knob"): L.mi_degensac_dev_set_knob(knob)
            best, mean, streamed, longest, busy, F, m, st = run(P, d, flags, tuning)
            same = ""
            if ref is not None: same = " identical: %s" % (np.array_equal(ref[0], F) and np.array_equal(ref[1], m) and np.array_equal(ref[2][:, :12], st[:, :12]))
            else: ref = (F, m, st)
            thr = st[:, 14]
            if len(set(thr.tolist())) > 1:          # a mixed-width launch: who ran what
                for t_ in sorted(set(thr.tolist())):
                    m_ = thr == t_; b_ = st[m_, 13].astype(np.float64) / 1e5; full_ = st[m_, 0] >= 99999
                    print(f"        {t_:4d} threads: pairs {int(m_.sum()):5d}  sum {b_.sum():9.1f} ms  mean {b_.mean():6.2f}  longest {b_.max():6.1f}  pairs with the whole budget {int(full_.sum())}  set aside {int(((st[m_, 15] >> 8) & 1).sum())}", flush=True)
            print(f"P={P:5d} {name:12s} best {best:7.2f} ms  mean {mean:7.2f} ms  streamed {streamed:4d}  longest pair {longest:6.1f} ms  sum of pair times {busy:9.1f} ms ({busy / 512:6.1f} per 512 slots; mean pair {busy / P:6.2f} ms){same}", flush=True)
