"""Where do the waves of co-resident workgroups sit?  Launches a grid shaped like the batch kernel's (workgroups of
`threads`, `dyn_lds` bytes of LDS so that two fit a CU) and prints, per CU, the SIMD of every wave of its workgroups."""
import os as _os
_os.environ.setdefault("MI_DEGENSAC_LIB", _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "libmi_degensac_dev.so"))
import sys, os, ctypes as C, numpy as np, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pydegensac_amd import _lib
L = _lib.lib()
threads, dyn = int(sys.argv[1]) if len(sys.argv) > 1 else 256, int(sys.argv[2]) if len(sys.argv) > 2 else 61440
grid = int(sys.argv[3]) if len(sys.argv) > 3 else 512
nw = threads // 64
o = np.zeros(grid * nw, np.int64)
rc = L.mi_degensac_hwid_probe(grid, threads, dyn, o.ctypes.data_as(C.POINTER(C.c_longlong)))
o = o.reshape(grid, nw)
hw = o & 0xffffffff; xcc = (o >> 32) & 0xf
simd = (hw >> 4) & 3; cu = (hw >> 8) & 15; sh = (hw >> 12) & 1; se = (hw >> 13) & 7
cuid = [(int(xcc[b, 0]), int(se[b, 0]), int(sh[b, 0]), int(cu[b, 0])) for b in range(grid)]
per = collections.defaultdict(list)
for b in range(grid): per[cuid[b]].append((b, tuple(int(x) for x in simd[b])))
print("CUs used", len(per), "workgroups per CU", collections.Counter(len(v) for v in per.values()))
same = sum(1 for v in per.values() if len(v) >= 2 and v[0][1][0] == v[1][1][0])
print("CUs whose two workgroups have wave 0 on the same SIMD:", same, "of", sum(1 for v in per.values() if len(v) >= 2))
pat = collections.Counter(tuple(w[1] for w in sorted(v)) for v in per.values())
for k, c in pat.most_common(8): print(c, k)
