import os as _os
# development exports live in libmi_degensac_dev.so (make -C pydegensac_amd/csrc dev), never in the product library
_os.environ.setdefault("MI_DEGENSAC_LIB", _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "libmi_degensac_loprof.so"))
import sys, numpy as np, ctypes as C, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pydegensac_amd import synthetic as syn, _lib, parallel
L=_lib.lib(); P=int(sys.argv[1]) if len(sys.argv)>1 else 256; N=int(sys.argv[2]) if len(sys.argv)>2 else 2000   # MI_DEGENSAC_TUNING=2 forces the 256-thread variant
IR=float(sys.argv[3]) if len(sys.argv)>3 else 0.4; MI=int(sys.argv[4]) if len(sys.argv)>4 else 100000
a=np.empty((P*N,2)); b=np.empty((P*N,2))
for i in range(P):
    p1,p2,_,_=syn.two_view_fundamental(N,IR,0.1,seed=i); a[i*N:(i+1)*N]=p1; b[i*N:(i+1)*N]=p2
offs=np.arange(P+1,dtype=np.int64)*N
dev=torch.device('cuda',0)
d_a=torch.from_numpy(a).to(dev); d_b=torch.from_numpy(b).to(dev); d_off=torch.from_numpy(offs).to(dev)
d_seeds=torch.from_numpy(parallel.pair_seeds(0,P).astype(np.int64)).to(dev).to(torch.int32)
d_F=torch.zeros((P,9),dtype=torch.float64,device=dev); d_mask=torch.zeros(P*N,dtype=torch.uint8,device=dev); d_st=torch.zeros((P,16),dtype=torch.int32,device=dev)
d_ph=torch.zeros((P,16),dtype=torch.int64,device=dev)
L.mi_degensac_debug_phases(C.c_void_p(d_ph.data_ptr()))
prm=_lib.make_params(0.5,0.9999,MI,0,True,0.0,True)
rc=L.mi_degensac_find_fundamental_batch_dev(d_a.data_ptr(),d_b.data_ptr(),d_off.data_ptr(),offs.ctypes.data_as(C.POINTER(C.c_int64)),P,2,C.byref(prm),d_seeds.data_ptr(),0,None,d_F.data_ptr(),d_mask.data_ptr(),d_st.data_ptr())
torch.cuda.synchronize()
ph=d_ph.cpu().numpy().astype(np.float64)/1e5; st=d_st.cpu().numpy()
names=["other","init pass","first fit","EX pass","Ss pass","hash||fit","commit","final pass","u2f14","-","reps taken from look-ahead (count x 1e5)","reps fitted (count x 1e5)"]
if os.environ.get("MI_DEGENSAC_TUNING", "0") in ("0", "1", "2", "3"):      # default = one repetition per wave (dg_inFrani_waves): its own timers
    names=["round: planning","round: repetitions (barrier to barrier)","round: replay","round: commit","round: generator / list restore","rounds (count x 1e5)","repetitions committed (count x 1e5)","between rounds",
           "wave 0: 14-point fit","wave 0: passes","wave 0: 8-point fits","wave 0: hash + lookup","wave 0: whole repetitions","wave 0: repetitions (count x 1e5)"]
lo=st[:,1].mean()
print("lo_runs mean",lo,"ex_passes mean",st[:,9].mean())
for i,nm in enumerate(names):
    if "count" in nm: print(f"{nm:46s} {ph[:,i].mean()*1e5/lo:6.2f} per LO run")
    else: print(f"{nm:12s} {ph[:,i].mean():8.3f} ms/pair   {ph[:,i].mean()/lo*1e3:8.1f} us/LO-run")
