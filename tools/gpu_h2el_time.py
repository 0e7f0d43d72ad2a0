import sys, time, numpy as np
sys.path.insert(0,'.')
import pydegensac_amd as pd
from pydegensac_amd import synthetic as syn
from oracle import ref
for n,ir in [(5000,0.4),(3000,0.1),(1000,0.3)]:
    u,_=syn.ellipse_pairs(n,ir,1.0,5,0.05)
    pd.ransacH2el(u,4.0,0.99,10000,True,0,seed=3)
    ts=[]
    for i in range(5):
        t=time.perf_counter(); H,m=pd.ransacH2el(u,4.0,0.99,10000,True,0,seed=3); ts.append(time.perf_counter()-t)
    st=pd.last_stats()
    print(n,ir,"gpu wall ms median %.2f"%(np.median(ts)*1e3),"kernel ms %.2f"%(st['ticks_total']/1e5),"samples",st['samples'],"lo",st['lo_runs'])
    U=[u]*64
    t=time.perf_counter(); pd.ransacH2el_batch(U,4.0,0.99,10000,True,0,seeds=list(range(1,65))); dt=time.perf_counter()-t
    print("   batch of 64: %.1f ms"%(dt*1e3))
