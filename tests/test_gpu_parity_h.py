"""GPU parity of the homography path (exp_ransacHcustomLAF): HIP kernel through the C-ABI vs the CPU oracle."""
import numpy as np
import pytest

import pydegensac_amd as pd
from pydegensac_amd import synthetic as syn
from tests import golden_util as gu

pytestmark = pytest.mark.gpu

CASES = [
    ("c3_sampson", dict(n=5000, inlier_ratio=0.4, sigma=0.5), dict(px_th=2.0)),
    ("c3_laf", dict(n=2000, inlier_ratio=0.4, sigma=0.5, laf=True), dict(px_th=2.0, laf_coef=3.0)),
    ("symm_sq_max", dict(n=1500, inlier_ratio=0.4, sigma=0.5), dict(px_th=2.0, error_type=1)),
    ("symm_max", dict(n=1500, inlier_ratio=0.4, sigma=0.5), dict(px_th=2.0, error_type=2)),
    ("symm_sq_sum", dict(n=1500, inlier_ratio=0.4, sigma=0.5), dict(px_th=2.0, error_type=3)),
    ("symm_sum", dict(n=1500, inlier_ratio=0.4, sigma=0.5), dict(px_th=2.0, error_type=4)),
    ("symm_max_laf", dict(n=1000, inlier_ratio=0.4, sigma=0.5, laf=True), dict(px_th=2.0, error_type=2, laf_coef=3.0)),
    ("symm_sum_laf", dict(n=1000, inlier_ratio=0.4, sigma=0.5, laf=True), dict(px_th=2.0, error_type=4, laf_coef=3.0)),
    ("low_inlier", dict(n=1000, inlier_ratio=0.1, sigma=0.5), dict(px_th=1.0)),
    ("nosym", dict(n=1000, inlier_ratio=0.3, sigma=0.5), dict(px_th=1.0, sym_check=False)),
    ("n4", dict(n=4, inlier_ratio=1.0, sigma=0.0), dict(px_th=1.0, max_iters=100)),
    ("all_outliers", dict(n=200, inlier_ratio=0.0, sigma=0.5), dict(px_th=1.0, max_iters=2000)),
]


@pytest.mark.parametrize("name,gen,kw", CASES, ids=[c[0] for c in CASES])
@pytest.mark.parametrize("seed", [1, 7])
def test_homography_matches_oracle(oracle_port, name, gen, kw, seed):
    p1, p2, _, _ = syn.homography_pairs(seed=2, **gen)
    Ho, mo, so = oracle_port.find_homography(p1, p2, seed=seed, **kw)
    H, m = pd.findHomography_(p1, p2, kw.get("px_th", 1.0), kw.get("conf", 0.999), kw.get("max_iters", 50000),
                              kw.get("error_type", 0), kw.get("sym_check", True), kw.get("laf_coef", 0.0), seed=seed)
    st = pd.last_stats()
    assert (st["samples"], st["lo_runs"], st["rejected"], st["models"]) == (so["samples"], so["lo_runs"], so["rejected"], so["models"])
    if np.abs(Ho).sum() == 0:
        assert np.abs(H).sum() == 0 and not np.asarray(m).any()
    else:
        assert np.array_equal(np.asarray(m), mo), f"{(np.asarray(m) != mo).sum()} mask bits differ"
        assert gu.rel(H, Ho) < 1e-6


def test_public_wrapper_inverts_like_reference(oracle_port):
    """findHomography returns inv(H.T) of the driver's model (utils.py:108) and maps image 1 -> image 2"""
    p1, p2, lab, Hgt = syn.homography_pairs(n=800, inlier_ratio=0.6, sigma=0.2, seed=5)
    H, mask = pd.findHomography(p1, p2, 1.0, seed=3)
    q = np.c_[p1[lab], np.ones(lab.sum())] @ H.T
    err = np.linalg.norm(q[:, :2] / q[:, 2:3] - p2[lab], axis=1)
    assert np.median(err) < 1.0 and np.asarray(mask)[lab].mean() > 0.9
