import sys, os, ctypes as C, numpy as np
sys.path.insert(0, os.getcwd())
from pydegensac_amd import _lib
L=_lib.lib(); o=np.zeros(2,np.int64)
print("rc", L.mi_degensac_atomic_order_probe(o.ctypes.data_as(C.POINTER(C.c_longlong))), "violations", o[0], "of", o[1])
