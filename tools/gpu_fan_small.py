"""One C2 pair per launch: the default configuration against the large-pair machinery (HBM placement, cooperative helpers, fan mode) forced
onto a 2000-correspondence pair.  usage: gpu_fan_small.py [pairs] [tuning words ...]"""
import sys, os, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pydegensac_amd import synthetic as syn, _lib, parallel
L = _lib.lib(); N = 2000; dev = torch.device('cuda', 0)
NP = int(sys.argv[1]) if len(sys.argv) > 1 else 32
TUNES = [int(x, 0) for x in sys.argv[2:]] or [0, 1 | (1 << 2) | (23 << 8)]
seeds_all = parallel.pair_seeds(0, NP).astype(np.int64)
def run(i, tune, reps=3):
    p1, p2, _, _ = syn.two_view_fundamental(N, 0.4, 0.1, seed=i)
    d_a = torch.from_numpy(p1).to(dev); d_b = torch.from_numpy(p2).to(dev); offs = np.array([0, N], np.int64); d_off = torch.from_numpy(offs).to(dev)
    d_seeds = torch.from_numpy(seeds_all[i:i+1]).to(dev).to(torch.int32)
    d_F = torch.zeros((1, 9), dtype=torch.float64, device=dev); d_mask = torch.zeros(N, dtype=torch.uint8, device=dev); d_st = torch.zeros((1, 16), dtype=torch.int32, device=dev)
    prm = _lib.make_params(0.5, 0.9999, 100000, 0, True, 0.0, True, 0, tune)
    ts = []
    for it in range(reps + 1):
        torch.cuda.synchronize(); t = time.perf_counter()
        _lib.check(L.mi_degensac_find_fundamental_batch_dev(d_a.data_ptr(), d_b.data_ptr(), d_off.data_ptr(), offs.ctypes.data_as(C.POINTER(C.c_int64)), 1, 2, C.byref(prm),
                                                            d_seeds.data_ptr(), 0, None, d_F.data_ptr(), d_mask.data_ptr(), d_st.data_ptr()))
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t) * 1e3)
    st = d_st.cpu().numpy()[0]
    return min(ts[1:]), st, d_F.cpu().numpy(), d_mask.cpu().numpy()
tot = {t: [] for t in TUNES}
for i in range(NP):
    ref = None; line = f"pair {i:3d}"
    for t in TUNES:
        ms, st, F, m = run(i, t)
        if ref is None: ref = (F, m, st[:12].copy()); line += f" samples {st[0]:6d} lo_runs {st[1]:2d}"
        same = np.array_equal(ref[0], F) and np.array_equal(ref[1], m) and np.array_equal(ref[2], st[:12])
        line += f" | tune {t:#x}: {ms:6.2f} ms thr {st[14]} plc {st[15] & 255:#x}{'' if same else ' DIFFERENT'}"
        tot[t].append(ms)
    print(line, flush=True)
for t in TUNES: print(f"tune {t:#x}: mean {np.mean(tot[t]):.2f} median {np.median(tot[t]):.2f} max {np.max(tot[t]):.2f}")
