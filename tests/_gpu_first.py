import sys, time, numpy as np
sys.path.insert(0, '.')
import pydegensac_amd as pd
from pydegensac_amd import synthetic as syn, _lib
from oracle import port
import ctypes as C
L = _lib.lib()
print(L.mi_degensac_version(), "devices", L.mi_degensac_device_count())
# 1. sample stream
n=2000; iters=600
out=np.zeros((iters,7),np.int32)
rc=L.mi_degensac_sample_stream(12345,n,7,iters,0,out.ctypes.data_as(C.POINTER(C.c_int32))); print("rc",rc, L.mi_degensac_last_error())
ref=np.zeros((iters,7),np.int32); port.lib().dg_oracle_sample_stream(12345,n,7,iters,port.ip(ref),None)
# oracle gives samidx order (reverse draw); device gives draw order
print("sample stream equal:", np.array_equal(out[:, ::-1], ref), (out[:, ::-1]!=ref).sum())
# 2. scoring
p1,p2,lab,Fgt = syn.two_view_fundamental(2000,0.4,0.1,seed=0)
rng=np.random.default_rng(0)
models=np.concatenate([Fgt.reshape(1,9), rng.normal(size=(31,9))]).copy()
I=np.zeros(32,np.uint32); J=np.zeros(32); res=np.zeros((32,n))
rc=L.mi_degensac_score_models(_lib.dptr(p1),_lib.dptr(p2),n,2,_lib.dptr(models),32,0,0.25,0,I.ctypes.data_as(C.POINTER(C.c_uint32)),_lib.dptr(J),_lib.dptr(res)); print("rc",rc,L.mi_degensac_last_error())
u=np.ones((n,6)); u[:,0:2]=p1; u[:,3:5]=p2
bad=0
for k in range(32):
    d=np.zeros(n); port.lib().dg_oracle_FDs(port.dp(u),port.dp(models[k].copy()),port.dp(d),n)
    inl=np.zeros(n,np.int32); S=port.lib().dg_oracle_inlidxs(port.dp(d),n,0.25,port.ip(inl))
    bad += (d!=res[k]).sum()
    if S.I!=I[k] or abs(S.J-J[k])>1e-9*max(1,abs(S.J)): print("score mismatch",k,S.I,I[k],S.J,J[k])
print("residual bit mismatches:", bad)
# 3. full runs
for n_,ir,sg,kw in [(500,0.5,0.1,dict(max_iters=20000)),(2000,0.4,0.1,{}),(2000,0.4,0.1,dict(pf=0.7,max_iters=3000))]:
    p1,p2,lab,_ = syn.two_view_fundamental(n_,ir,sg,seed=3,plane_fraction=kw.pop('pf',0.0))
    for seed in [1,7]:
        t=time.perf_counter(); F,m = pd.findFundamentalMatrix_(p1,p2,0.5,0.9999,kw.get('max_iters',100000),0,True,0.0,True,seed=seed); dt=time.perf_counter()-t
        st=pd.last_stats()
        Fo,mo,so = port.find_fundamental(p1,p2,seed=seed,max_iters=kw.get('max_iters',100000))
        a=F/np.linalg.norm(F); b=Fo/np.linalg.norm(Fo)
        print(n_,seed,"GPU",{k:st[k] for k in ['samples','lo_runs','degen','Ih','full_passes','ex_passes','h_passes','aux_passes','ticks_total']}, m.sum(), f"{dt*1e3:.1f}ms")
        print("      ORA",{k:so[k] for k in ['samples','lo_runs','degen','Ih','full_passes','ex_passes','hds_passes','fds_direct']}, mo.sum(), "maskdiff",(m!=mo).sum(),"relF",np.linalg.norm(a-b))
