"""Matcher stage (examples/simple-example.py:46-53): the numpy oracle against a float64 brute force on CPU,
and the HIP kernels (mi_degensac_match*) against the oracle on the GPU — indices, distances and ratio / mutual decisions
bit for bit, including exact ties, duplicates, ragged tile sizes and degenerate train sets."""
import numpy as np
import pytest

from oracle import matcher_np as mo


def _descs(rng, n1, n2, dim, kind):
    if kind == "l2":
        b = rng.normal(size=(n2, dim)).astype(np.float32)
        a = rng.normal(size=(n1, dim)).astype(np.float32)
        m = min(n1, n2) // 2
        a[:m] = b[rng.permutation(n2)[:m]] + 0.05 * rng.normal(size=(m, dim)).astype(np.float32)   # true matches
        if n2 > 8:
            b[5] = b[3]; b[7] = b[3]                                                               # exact duplicates: ties
        return a, b
    b = rng.integers(0, 256, size=(n2, dim), dtype=np.uint8)
    a = rng.integers(0, 256, size=(n1, dim), dtype=np.uint8)
    m = min(n1, n2) // 2
    a[:m] = b[rng.permutation(n2)[:m]] ^ (rng.random((m, dim)) < 0.03).astype(np.uint8)
    if n2 > 8:
        b[5] = b[3]
    return a, b


def test_oracle_agrees_with_float64_bruteforce():
    rng = np.random.default_rng(0)
    a, b = _descs(rng, 200, 300, 64, "l2")
    idx, dist = mo.knn2(a, b, "l2")
    D = np.sqrt(((a[:, None, :].astype(np.float64) - b[None, :, :].astype(np.float64)) ** 2).sum(-1))
    nn = np.argsort(D, axis=1, kind="stable")[:, :2]
    gap = np.sort(D, axis=1)[:, 1:3]
    clear = (gap[:, 1] - gap[:, 0] > 1e-4) & (np.sort(D, axis=1)[:, 1] - np.sort(D, axis=1)[:, 0] > 1e-4)
    assert np.array_equal(idx[clear], nn[clear]) and clear.mean() > 0.9
    assert np.allclose(dist, np.take_along_axis(D, idx.astype(np.int64), 1), rtol=1e-5)
    a, b = _descs(rng, 100, 150, 61, "hamming")
    idx, dist = mo.knn2(a, b, "hamming")
    H = np.array([[bin(int.from_bytes(bytes(x ^ y), "little")).count("1") for y in b] for x in a], np.float32)
    assert np.array_equal(dist[:, 0], H.min(axis=1)) and np.array_equal(idx[:, 0], H.argmin(axis=1))
    q, t, d = mo.match_snn(a, b, 0.9, mutual=True, norm="hamming")
    assert len(q) > 20 and np.array_equal(H[q, t], d)


@pytest.mark.gpu
@pytest.mark.parametrize("kind,dim", [("l2", 64), ("l2", 128), ("l2", 37), ("hamming", 61), ("hamming", 32)])
@pytest.mark.parametrize("n1,n2", [(1500, 1700), (64, 64), (1, 3), (257, 1), (130, 2)])
def test_gpu_matcher_equals_oracle(kind, dim, n1, n2):
    from pydegensac_amd import matcher
    rng = np.random.default_rng(n1 * 7 + n2 + dim)
    a, b = _descs(rng, n1, n2, dim, kind)
    idx, dist = matcher.knn_match(a, b, kind)
    ri, rd = mo.knn2(a, b, kind)
    assert np.array_equal(idx, ri)
    assert np.array_equal(dist, rd)                                   # bit for bit (inf where there is no second row)
    for mutual in (False, True):
        q, t, d = matcher.match_snn(a, b, 0.9, mutual, kind)
        rq, rt, rdd = mo.match_snn(a, b, 0.9, mutual, kind)
        assert np.array_equal(q, rq) and np.array_equal(t, rt) and np.array_equal(d, rdd)


@pytest.mark.gpu
def test_gpu_example_pipeline_from_descriptors_to_homography(oracle_port):
    """the reference example's call pattern (simple-example.py:46-53 then verify_pydegensac, :18-23) with the matcher on
    the GPU: synthetic keypoints + descriptors of two views related by a homography"""
    import pydegensac
    from pydegensac_amd import matcher, synthetic as syn
    rng = np.random.default_rng(3)
    p1, p2, lab, Hgt = syn.homography_pairs(n=1200, inlier_ratio=0.5, sigma=0.5, seed=11)
    d1 = rng.normal(size=(1200, 64)).astype(np.float32)
    d2 = d1 + 0.15 * rng.normal(size=d1.shape).astype(np.float32)
    d2[~lab] = rng.normal(size=((~lab).sum(), 64)).astype(np.float32)          # outliers: unrelated descriptors
    perm = rng.permutation(1200); kps2 = p2[perm]; descs2 = d2[perm]
    src, dst = matcher.tentative_points(p1, kps2, d1, descs2, 0.9)
    q, t, _ = mo.match_snn(d1, descs2, 0.9)
    assert np.array_equal(src, p1[q]) and np.array_equal(dst, kps2[t]) and len(q) > 400
    H, mask = pydegensac.findHomography(src, dst, 4.0, 0.99, 2000, seed=2)
    assert np.asarray(mask).sum() > 0.8 * len(q)


@pytest.mark.gpu
def test_gpu_keypoints_to_laf_rows_equal_the_reference_conversion():
    """mi_degensac_kpts_to_xyA against utils.py:24-41 (pydegensac_amd.convert_cv2_kpts_to_xyA on keypoint objects)"""
    import math
    import pydegensac_amd as pd
    from pydegensac_amd import matcher

    class KP:                                                     # what utils.py reads of a cv2.KeyPoint
        def __init__(self, x, y, s, a): self.pt = (x, y); self.size = s; self.angle = a
    rng = np.random.default_rng(5)
    k = np.c_[rng.uniform(0, 800, 500), rng.uniform(0, 600, 500), rng.uniform(2, 60, 500), rng.uniform(0, 360, 500)].astype(np.float32)
    k[0, 3] = 0.0; k[1, 3] = 90.0; k[2, 3] = 180.0
    want = pd.convert_cv2_kpts_to_xyA([KP(float(a), float(b), float(c), float(d)) for a, b, c, d in k])
    got = matcher.kpts_to_xyA(k)
    assert np.array_equal(got[:, :2], want[:, :2])
    assert np.allclose(got, want, rtol=0, atol=4e-14 * 60)         # cos / sin of the device library vs libm: last-bit agreement
