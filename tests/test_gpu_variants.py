"""GPU: every kernel variant (512 / 256 threads per workgroup) and placement mode (0: points + pool in the HBM
workspace, 1: both in LDS, 2: pool in LDS) returns the same bits, and the oracle agrees with them."""
import os

import numpy as np
import pytest

import pydegensac_amd as pd
from pydegensac_amd import synthetic as syn

pytestmark = pytest.mark.gpu


@pytest.fixture
def force(monkeypatch):
    def _set(variant, mode):
        monkeypatch.setenv("MI_DEGENSAC_VARIANT", str(variant)); monkeypatch.setenv("MI_DEGENSAC_MODE", str(mode))
    yield _set


def _f_batch():
    A, B = [], []
    for i, (n, pf) in enumerate([(2000, 0.0), (900, 0.0), (1500, 0.7), (300, 0.0)]):
        p1, p2, _, _ = syn.two_view_fundamental(n, 0.4, 0.1, seed=20 + i, plane_fraction=pf); A.append(p1); B.append(p2)
    return A, B


def test_fundamental_variants_and_modes_agree(force, oracle_port):
    A, B = _f_batch(); seeds = [1, 2, 3, 4]
    ref = None
    for variant in (512, 256):
        for mode in (1, 2, 0):
            force(variant, mode)
            F, m = pd.findFundamentalMatrixBatch(A, B, max_iters=20000, seeds=seeds)
            if ref is None:
                ref = (np.asarray(F).copy(), [np.asarray(x).copy() for x in m])
            else:
                assert np.array_equal(np.asarray(F), ref[0]), (variant, mode)
                assert all(np.array_equal(np.asarray(x), y) for x, y in zip(m, ref[1])), (variant, mode)
    for p in range(len(A)):
        Fo, mo, _ = oracle_port.find_fundamental(A[p], B[p], 0.5, 0.9999, 20000, seed=seeds[p])
        assert np.array_equal(ref[1][p], mo.astype(bool))
        a = ref[0][p].ravel(); b = np.asarray(Fo).ravel()
        assert np.linalg.norm(a - b) <= 1e-9 * np.linalg.norm(b)


def test_homography_variants_and_modes_agree(force):
    A, B = [], []
    for i, n in enumerate([1200, 400, 2500]):
        p1, p2, _, _ = syn.homography_pairs(n, 0.4, 0.5, seed=30 + i, laf=True); A.append(p1); B.append(p2)
    ref = None
    for variant in (512, 256):
        for mode in (1, 2, 0):
            force(variant, mode)
            H, m = pd.findHomographyBatch(A, B, 1.0, 0.999, 20000, 3.0, "sampson", True, seeds=[5, 6, 7])
            if ref is None:
                ref = (np.asarray(H).copy(), [np.asarray(x).copy() for x in m])
            else:
                assert np.array_equal(np.asarray(H), ref[0]), (variant, mode)
                assert all(np.array_equal(np.asarray(x), y) for x, y in zip(m, ref[1])), (variant, mode)
