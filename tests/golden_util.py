import ast
import glob
import os

import numpy as np

from pydegensac_amd import synthetic as syn

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def fixtures(kind):
    return sorted(f for f in glob.glob(os.path.join(HERE, f"{kind}_*.npz")))        # "L" = L_*.npz only, "LS" = LS_*.npz


def load(path):
    z = np.load(path, allow_pickle=False)
    g = ast.literal_eval(str(z["gen"])); call = ast.literal_eval(str(z["call"]))
    kind = str(z["kind"])
    if kind in ("F", "L"):
        p1, p2, _, _ = syn.two_view_fundamental(**g)
    elif kind == "E":
        p1, _ = syn.ellipse_pairs(**g); p2 = None          # p1 = u10 [n, 10]
    else:
        p1, p2, _, _ = syn.homography_pairs(**g)
    n = int(z["n"])
    mask = np.unpackbits(z["mask"])[:n].astype(bool)
    return dict(kind=kind, variant=int(z["variant"]) if "variant" in z else None, n=n, p1=p1, p2=p2, call=call, seed=int(z["seed"]), model=z["model"], mask=mask,
                samples=int(z["samples"]), lo_runs=int(z["lo_runs"]), full_passes=int(z["full_passes"]),
                ex_passes=int(z["ex_passes"]) if "ex_passes" in z else None,
                rejected=int(z["rejected"]) if "rejected" in z else None, I=int(z["I"]))


def rel(a, b):
    a = np.asarray(a, float).ravel(); b = np.asarray(b, float).ravel()
    na, nb = np.linalg.norm(a), np.linalg.norm(b)
    if na == 0 or nb == 0:
        return 0.0 if na == nb else 1.0
    return np.linalg.norm(a / na - b / nb)
