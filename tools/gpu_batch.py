"""Batch throughput probe: python tools/gpu_batch.py P [tuning]  -> ms per batch, models/s."""
import sys, numpy as np, ctypes as C, torch, time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pydegensac_amd import synthetic as syn, _lib, parallel
L=_lib.lib()
P=int(sys.argv[1]) if len(sys.argv)>1 else 512
N=2000; U=min(P,512)
a=np.empty((P*N,2)); b=np.empty((P*N,2))
for i in range(U):
    p1,p2,_,_=syn.two_view_fundamental(N,0.4,0.1,seed=i); a[i*N:(i+1)*N]=p1; b[i*N:(i+1)*N]=p2
for i in range(U,P): a[i*N:(i+1)*N]=a[(i%U)*N:(i%U+1)*N]; b[i*N:(i+1)*N]=b[(i%U)*N:(i%U+1)*N]
offs=np.arange(P+1,dtype=np.int64)*N
dev=torch.device('cuda',0)
d_a=torch.from_numpy(a).to(dev); d_b=torch.from_numpy(b).to(dev); d_off=torch.from_numpy(offs).to(dev)
d_seeds=torch.from_numpy(parallel.pair_seeds(0,P).astype(np.int64)).to(dev).to(torch.int32)
d_F=torch.zeros((P,9),dtype=torch.float64,device=dev); d_mask=torch.zeros(P*N,dtype=torch.uint8,device=dev); d_st=torch.zeros((P,16),dtype=torch.int32,device=dev)
TUNE=int(sys.argv[2],0) if len(sys.argv)>2 else 0      # params.tuning (include/mi_degensac.h): 1 = latency variant, 2 = throughput
prm=_lib.make_params(0.5,0.9999,100000,0,True,0.0,True,0,TUNE)
best=1e9
for it in range(3):
    torch.cuda.synchronize(); t=time.perf_counter()
    rc=L.mi_degensac_find_fundamental_batch_dev(d_a.data_ptr(),d_b.data_ptr(),d_off.data_ptr(),offs.ctypes.data_as(C.POINTER(C.c_int64)),P,2,C.byref(prm),d_seeds.data_ptr(),0,None,d_F.data_ptr(),d_mask.data_ptr(),d_st.data_ptr())
    torch.cuda.synchronize(); dt=time.perf_counter()-t; best=min(best,dt)
st=d_st.cpu().numpy()
print(f"variant={st[0,14]} mode={st[0,15]} P={P} rc={rc} batch_ms={best*1e3:.1f} models/s={st[:,4].sum()/best/1e6:.2f}M pairs/s={P/best:.0f} sumI={st[:,3].sum()} ticks_total mean ms={st[:,13].mean()/1e5:.2f} max={st[:,13].max()/1e5:.1f}")
