import os as _os
# development exports live in libmi_degensac_dev.so (make -C pydegensac_amd/csrc dev), never in the product library
_os.environ.setdefault("MI_DEGENSAC_LIB", _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "libmi_degensac_dev.so"))
import sys, os, ctypes as C, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pydegensac_amd import _lib
L=_lib.lib(); o=np.zeros(16,np.int64)
L.mi_degensac_latency_probe(o.ctypes.data_as(C.POINTER(C.c_longlong)))
names=["fma","mul","add","div+add","sqrt+add","lartg_fast+2","lartg+2","readlane(dyn)+add","u32 mad","(clock)","lartg_bf+2"]
for n,v in zip(names,o): print(f"{n:20s} {v*10/4000:.1f} ns per op")
print("fma chain clock64 cycles per op:", o[9]/4000, " => wall/clock ratio: ns per clock64 tick", (o[0]*10)/max(o[9],1))

print("raw ticks (10 ns):", list(o))
print("fp64 fma x16000: dependent chain, 64 lanes %.2f ns/op; 16 lanes %.2f; 1 lane %.2f; 4 independent chains %.2f ns/op" % tuple(o[k]*10/16000 for k in (11,12,13,14)))
