"""One C2 pair through the host-staging API (no torch): smoke / hang probe."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import pydegensac_amd as pd
from pydegensac_amd import synthetic as syn
p1,p2,lab,F=syn.two_view_fundamental(2000,0.4,0.1,seed=3)
t=time.time(); Fm,mask=pd.findFundamentalMatrix(p1,p2,0.5,0.9999,100000,seed=1); print("ok F", int(mask.sum()), round(time.time()-t,3), os.environ.get("MI_DEGENSAC_LIB","default"), flush=True)
