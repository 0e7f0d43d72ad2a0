import os as _os
# development exports live in libmi_degensac_dev.so (make -C pydegensac_amd/csrc dev), never in the product library
_os.environ.setdefault("MI_DEGENSAC_LIB", _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "libmi_degensac_dev.so"))
import sys, numpy as np, ctypes as C
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pydegensac_amd import synthetic as syn, _lib
L=_lib.lib()
p1,p2,lab,F=syn.two_view_fundamental(2000,1.0,0.1,seed=0)
inp=np.zeros(512)
inp[:64]=np.c_[p1[:16],p2[:16]].ravel()
rng=np.random.default_rng(0); inp[64:64+14*9]=rng.normal(size=14*9); inp[200:209]=F.ravel()/np.linalg.norm(F)
t=np.zeros(8,np.int64); reps=50
mo=np.zeros(128); L.mi_degensac_microbench_out(mo.ctypes.data_as(C.POINTER(C.c_double)))
rc=L.mi_degensac_microbench(inp.ctypes.data_as(C.POINTER(C.c_double)),reps,t.ctypes.data_as(C.POINTER(C.c_longlong)))
names=["cov9+eig9 WAVE","cov9+eig9 lane0","u2f_small(14)","u2f_small(8)","singulF","checksample","u2h_small(5)","hash(800)"]
for n,v in zip(names,t):
    if n=="checksample": print(f"lartg_fast mismatches (of 1.28M): {v}")
    elif n=="cov9+eig9 lane0": print(f"lartg_bf mismatches (of 1.28M): {v}")
    else: print(f"{n:16s} {v/reps/100:.1f} us")

if os.environ.get("MI_DEGENSAC_LIB","").endswith("exp_et.so"): print("steqr split (us per call): outside the rotation chains %.1f, inside %.1f" % (mo[12]/reps/100, mo[13]/reps/100)); print("eig stages (us per call): tridiag %.1f  orgtr %.1f  steqr %.1f  sort %.1f" % tuple(t[4:8]/reps/100))

for k in range(8):
    o=mo[16+8*k:24+8*k]
    if o[0]==0 and o[1]==0: break
    print("lartg mismatch f=%r g=%r | plain c,s,r = %r %r %r | fast = %r %r %r" % tuple(o))
