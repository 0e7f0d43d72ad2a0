import os as _os
# development exports live in libmi_degensac_dev.so (make -C pydegensac_amd/csrc dev), never in the product library
_os.environ.setdefault("MI_DEGENSAC_LIB", _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "libmi_degensac_dev.so"))
import sys, os, ctypes as C, numpy as np
sys.path.insert(0, os.getcwd())
from pydegensac_amd import _lib
L=_lib.lib(); o=np.zeros(2,np.int64)
print("rc", L.mi_degensac_atomic_order_probe(o.ctypes.data_as(C.POINTER(C.c_longlong))), "violations", o[0], "of", o[1])
