"""Run golden F fixtures one by one in this process: python tools/gpu_one.py name [name...]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import pydegensac_amd as pd
from tests import golden_util as gu
for name in sys.argv[1:]:
    g = gu.load(os.path.join(os.path.dirname(gu.__file__), "golden", name + ".npz")); kw = g["call"]
    F, m = pd.findFundamentalMatrix_(g["p1"], g["p2"], kw.get("px_th", 0.5), kw.get("conf", 0.9999), kw.get("max_iters", 100000),
                                     kw.get("error_type", 0), kw.get("sym_check", True), kw.get("laf_coef", 0.0), kw.get("degen", True), seed=g["seed"])
    st = pd.last_stats()
    print(name, "gpu", st["samples"], st["lo_runs"], st["full_passes"], st["ex_passes"], "golden", g["samples"], g["lo_runs"], g["full_passes"], g["ex_passes"], "mask_eq", bool(np.array_equal(np.asarray(m), g["mask"])))
