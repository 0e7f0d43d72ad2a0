import os as _os
# development exports live in libmi_degensac_dev.so (make -C pydegensac_amd/csrc dev), never in the product library
_os.environ.setdefault("MI_DEGENSAC_LIB", _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "libmi_degensac_dev.so"))
import sys, numpy as np, ctypes as C, torch, time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pydegensac_amd import synthetic as syn, _lib, parallel
L=_lib.lib()
P=int(sys.argv[1]) if len(sys.argv)>1 else 256
N=int(sys.argv[2]) if len(sys.argv)>2 else 2000
IR=float(sys.argv[3]) if len(sys.argv)>3 else 0.4
MI=int(sys.argv[4]) if len(sys.argv)>4 else 100000
I0=int(os.environ.get('PH_FIRST','0'))   # first pair (data seed and pair seed) of the batch
TUNE=int(os.environ.get('PH_TUNE','0'),0)
a=np.empty((P*N,2)); b=np.empty((P*N,2))
for i in range(P):
    p1,p2,_,_=syn.two_view_fundamental(N,IR,0.1,seed=I0+i); a[i*N:(i+1)*N]=p1; b[i*N:(i+1)*N]=p2
offs=np.arange(P+1,dtype=np.int64)*N
dev=torch.device('cuda',0)
d_a=torch.from_numpy(a).to(dev); d_b=torch.from_numpy(b).to(dev); d_off=torch.from_numpy(offs).to(dev)
d_seeds=torch.from_numpy(parallel.pair_seeds(0,I0+P)[I0:].astype(np.int64)).to(dev).to(torch.int32)
d_F=torch.zeros((P,9),dtype=torch.float64,device=dev); d_mask=torch.zeros(P*N,dtype=torch.uint8,device=dev); d_st=torch.zeros((P,16),dtype=torch.int32,device=dev)
d_ph=torch.zeros((2*P+4096,16),dtype=torch.int64,device=dev)   # + the workgroups' exit times and the set-aside records of the dev build
L.mi_degensac_debug_phases(C.c_void_p(d_ph.data_ptr()))
prm=_lib.make_params(0.5,0.9999,MI,0,True,0.0,True,0,TUNE)
for it in range(2):
    torch.cuda.synchronize(); t=time.perf_counter()
    rc=L.mi_degensac_find_fundamental_batch_dev(d_a.data_ptr(),d_b.data_ptr(),d_off.data_ptr(),offs.ctypes.data_as(C.POINTER(C.c_int64)),P,2,C.byref(prm),d_seeds.data_ptr(),0,None,d_F.data_ptr(),d_mask.data_ptr(),d_st.data_ptr())
    torch.cuda.synchronize(); dt=time.perf_counter()-t
print("rc",rc,"batch ms",dt*1e3)
ph=d_ph.cpu().numpy()[:P].astype(np.float64)/1e5  # ms
st=d_st.cpu().numpy()
names=["sample","solve","score","commit+misc","LO","degen","tail","total","innerH","rFtH_gen","rFtH_score","rFtH_trig","chain","draws","pool","d7"]
print("mean ms per pair:", {n:round(float(ph[:,i].mean()),3) for i,n in enumerate(names)})
print("max  ms per pair:", {n:round(float(ph[:,i].max()),3) for i,n in enumerate(names)})
print("samples mean",st[:,0].mean(),"lo_runs mean",st[:,1].mean(),"degen mean",st[:,5].mean(),"models mean",st[:,4].mean(), "aux", st[:,11].mean(), "hds", st[:,10].mean())
worst=np.argsort(-ph[:,7])[:5]
for w in worst: print("pair",w,{n:round(float(ph[w,i]),1) for i,n in enumerate(names)},"stats",list(st[w,:12]))
