import sys, numpy as np, time
sys.path.insert(0,'.')
import pydegensac_amd as pd
from pydegensac_amd import synthetic as syn
from oracle import port
bad=0
for n,ir,seed,laf,et in [(70000,0.3,5,False,0),(70000,0.2,6,True,2),(100000,0.5,7,False,0),(66000,0.1,8,True,0)]:
    p1,p2,_,_=syn.homography_pairs(n,ir,0.5,seed=seed,laf=laf)
    for tn in (0,1,3|(1<<2)):
        t=time.time(); H,m=pd.findHomography_(p1,p2,2.0,0.999,20000,et,True,3.0 if laf else 0.0,seed=seed,tuning=tn); dt=time.time()-t; st=pd.last_stats()
        Ho,mo,so=port.find_homography(p1,p2,2.0,0.999,20000,et,True,3.0 if laf else 0.0,seed=seed)
        ok=np.array_equal(np.asarray(m,bool),mo) and (st['samples'],st['lo_runs'],st['models'])==(so['samples'],so['lo_runs'],so['models'])
        rel=np.linalg.norm(np.asarray(H).ravel()-Ho.ravel())/np.linalg.norm(Ho.ravel())
        print(n,ir,laf,et,tn,"ok" if ok and rel<1e-9 else "MISMATCH",st['samples'],st['lo_runs'],st['threads'],st['placement'],"%.1f ms"%(dt*1e3),rel)
        bad+= not (ok and rel<1e-9)
print("bad",bad)
