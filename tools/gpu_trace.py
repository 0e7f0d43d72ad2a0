import os as _os
# development exports live in libmi_degensac_dev.so (make -C pydegensac_amd/csrc dev), never in the product library
_os.environ.setdefault("MI_DEGENSAC_LIB", _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "libmi_degensac_dev.so"))
import sys, numpy as np, ctypes as C
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pydegensac_amd as pd
from pydegensac_amd import synthetic as syn, _lib
from oracle import port
L=_lib.lib(); P=port.lib()
dseed=int(sys.argv[2]) if len(sys.argv)>2 else 3
p1,p2,lab,_ = syn.two_view_fundamental(2000,0.4,0.1,seed=dseed)
seed=int(sys.argv[1]) if len(sys.argv)>1 else 1
tr_o=[]
CB=C.CFUNCTYPE(None,C.c_int,C.c_int,C.c_double)
cb=CB(lambda t,i,j: tr_o.append((t,i,j)))
P.dg_oracle_set_trace2(cb)
Fo,mo,so=port.find_fundamental(p1,p2,seed=seed)
P.dg_oracle_set_trace2(CB(0))
cap=20000
L.mi_degensac_debug_trace(cap,None)
F,m=pd.findFundamentalMatrix_(p1,p2,0.5,0.9999,100000,0,True,0.0,True,seed=seed)
buf=np.zeros(1+4*cap,np.int32); L.mi_degensac_debug_trace(0,buf.ctypes.data_as(C.POINTER(C.c_int)))
k=buf[0]; rec=buf[1:1+4*k].reshape(k,4)
tr_g=[(int(r[0]),int(r[1]),float(np.array([(int(r[2])&0xffffffff)|(int(r[3])<<32)],dtype=np.int64).view(np.float64)[0])) for r in rec]
print(len(tr_o),len(tr_g))
for i,(a,b) in enumerate(zip(tr_o,tr_g)):
    if a[0]==13 and b[0]==13: continue
    if a[0]!=b[0] or a[1]!=b[1] or abs(a[2]-b[2])>1e-9*max(1,abs(a[2])):
        print("first diff at",i)
        for j in range(max(0,i-6),min(len(tr_o),len(tr_g),i+6)): print(j,tr_o[j],tr_g[j])
        break
else: print("traces equal over common prefix")
