"""GPU parity of the fundamental-matrix path: the HIP kernel (through the C-ABI) against the CPU
oracle on the same seeded inputs.  Integer outputs (mask, sample / LO / model counts) bit-exact;
F within 1e-6 relative Frobenius (north_star tolerance)."""
import numpy as np
import pytest

import pydegensac_amd as pd
from pydegensac_amd import synthetic as syn

pytestmark = pytest.mark.gpu


def relF(a, b):
    a = a / np.linalg.norm(a); b = b / np.linalg.norm(b)
    return np.linalg.norm(a - b)


def run_pair(port, p1, p2, seed, **kw):
    kwo = dict(px_th=kw.get("px_th", 0.5), conf=kw.get("conf", 0.9999), max_iters=kw.get("max_iters", 100000),
               error_type=kw.get("error_type", 0), sym_check=kw.get("sym_check", True), degen=kw.get("degen", True),
               laf_coef=kw.get("laf_coef", 0.0))
    Fo, mo, so = port.find_fundamental(p1, p2, seed=seed, **kwo)
    F, m = pd.findFundamentalMatrix_(p1, p2, kwo["px_th"], kwo["conf"], kwo["max_iters"], kwo["error_type"],
                                     kwo["sym_check"], kwo["laf_coef"], kwo["degen"], seed=seed)
    st = pd.last_stats()
    return (F, m, st), (Fo, mo, so)


CASES = [
    ("c2", dict(n=2000, ir=0.4, sigma=0.1), {}),
    ("c2b_plane", dict(n=2000, ir=0.4, sigma=0.1, plane=0.7), dict(max_iters=3000)),
    ("n500", dict(n=500, ir=0.5, sigma=0.1), dict(max_iters=20000)),
    ("n100_low", dict(n=100, ir=0.3, sigma=0.3), dict(max_iters=5000)),
    ("n20", dict(n=20, ir=0.8, sigma=0.1), dict(max_iters=2000)),
    ("n8", dict(n=8, ir=1.0, sigma=0.1), dict(max_iters=200)),
    ("symm_epipolar", dict(n=1000, ir=0.4, sigma=0.1), dict(error_type=1)),
    ("nodegen", dict(n=1000, ir=0.4, sigma=0.1, plane=0.7), dict(degen=False)),
    ("nosym", dict(n=1000, ir=0.4, sigma=0.3), dict(sym_check=False)),
    ("noise_at_threshold", dict(n=1000, ir=0.4, sigma=0.5), dict(max_iters=8000)),
    ("all_outliers", dict(n=300, ir=0.0, sigma=0.5), dict(max_iters=3000)),
]


@pytest.mark.parametrize("name,gen,kw", CASES, ids=[c[0] for c in CASES])
@pytest.mark.parametrize("seed", [1, 7])
def test_fundamental_matches_oracle(oracle_port, name, gen, kw, seed):
    p1, p2, lab, _ = syn.two_view_fundamental(gen["n"], gen["ir"], gen["sigma"], seed=3, plane_fraction=gen.get("plane", 0.0))
    (F, m, st), (Fo, mo, so) = run_pair(oracle_port, p1, p2, seed, **kw)
    assert st["samples"] == so["samples"]
    assert st["lo_runs"] == so["lo_runs"]
    assert st["degen"] == so["degen"]
    assert st["full_passes"] == so["full_passes"] and st["ex_passes"] == so["ex_passes"]
    assert np.array_equal(np.asarray(m), mo), f"{(np.asarray(m) != mo).sum()} mask bits differ"
    if np.abs(Fo).sum() == 0:
        assert np.abs(F).sum() == 0
    else:
        assert relF(F, Fo) < 1e-6
