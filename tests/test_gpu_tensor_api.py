"""GPU: the device-resident batch API returns exactly what the host-staging API returns."""
import numpy as np
import pytest

import pydegensac_amd as pd
from pydegensac_amd import synthetic as syn, tensor_api

pytestmark = pytest.mark.gpu


def _pairs(kind, sizes):
    A, B = [], []
    for i, n in enumerate(sizes):
        if kind == "F":
            p1, p2, _, _ = syn.two_view_fundamental(n, 0.4, 0.1, seed=10 + i)
        else:
            p1, p2, _, _ = syn.homography_pairs(n, 0.4, 0.5, seed=10 + i, laf=True)
        A.append(p1); B.append(p2)
    return A, B


def test_fundamental_tensors_match_host_batch(oracle_port):
    import torch
    sizes = [700, 1200, 64, 2000]                                     # ragged batch
    A, B = _pairs("F", sizes); seeds = [3, 5, 7, 11]
    Fh, mh = pd.findFundamentalMatrixBatch(A, B, seeds=seeds)
    dev = torch.device("cuda", 0)
    F, m, st, offs = tensor_api.find_fundamental_batch_tensors(torch.from_numpy(np.concatenate(A)).to(dev),
                                                               torch.from_numpy(np.concatenate(B)).to(dev), sizes, seeds=seeds)
    assert F.is_cuda and m.is_cuda and st.is_cuda and F.shape == (4, 3, 3) and m.dtype == torch.bool
    F = F.cpu().numpy(); m = m.cpu().numpy()
    assert np.array_equal(F, np.asarray(Fh))
    for p in range(4):
        assert np.array_equal(m[offs[p]:offs[p + 1]], np.asarray(mh[p]))
    stn = st.cpu().numpy()
    assert (stn[:, 0] > 0).all()                          # samples drawn
    for p in range(4):                                     # ... and both are the CPU oracle's results
        Fo, mo, so = oracle_port.find_fundamental(A[p], B[p], 0.5, 0.9999, 100000, seed=seeds[p])
        assert (int(stn[p, 0]), int(stn[p, 1])) == (so["samples"], so["lo_runs"]), p
        assert np.array_equal(m[offs[p]:offs[p + 1]], mo), p
        assert np.linalg.norm(F[p].ravel() - np.asarray(Fo).ravel()) <= 1e-9 * np.linalg.norm(Fo), p


def test_homography_tensors_match_host_batch(oracle_port):
    import torch
    sizes = [900, 300, 1500]
    A, B = _pairs("H", sizes); seeds = [2, 4, 6]
    Hh, mh = pd.findHomographyBatch(A, B, 1.0, 0.999, 20000, 3.0, "symm_max", True, seeds=seeds)
    dev = torch.device("cuda", 0)
    H, m, st, offs = tensor_api.find_homography_batch_tensors(torch.from_numpy(np.concatenate(A)).to(dev),
                                                              torch.from_numpy(np.concatenate(B)).to(dev), sizes, 1.0, 0.999, 20000,
                                                              3.0, "symm_max", True, seeds=seeds)
    H = H.cpu().numpy(); m = m.cpu().numpy(); st = st.cpu().numpy()
    for p in range(3):
        assert np.array_equal(m[offs[p]:offs[p + 1]], np.asarray(mh[p]))
        # inv() runs in numpy on the host path and in torch.linalg on the device path: same to rounding
        assert np.linalg.norm(H[p] - Hh[p]) <= 1e-9 * max(np.linalg.norm(Hh[p]), 1e-300)
        Ho, mo, so = oracle_port.find_homography(A[p], B[p], 1.0, 0.999, 20000, 2, True, 3.0, seed=seeds[p])      # "symm_max" = error type 2
        assert (int(st[p, 0]), int(st[p, 1])) == (so["samples"], so["lo_runs"]), p
        assert np.array_equal(m[offs[p]:offs[p + 1]], mo), p
        if np.abs(Ho).sum() > 0:
            Hu = np.linalg.inv(np.asarray(Ho).T)
            assert np.linalg.norm(H[p] - Hu) <= 1e-8 * np.linalg.norm(Hu), p


def test_device_pipeline_descriptors_to_homography_matches_host_stages():
    """keypoints + descriptors on the device -> matcher -> LAF rows -> findHomography, nothing staged through the host but
    the survivor count; every stage equals its host-API counterpart"""
    import torch
    from pydegensac_amd import matcher
    rng = np.random.default_rng(9)
    p1, p2, lab, _ = syn.homography_pairs(n=1500, inlier_ratio=0.5, sigma=0.5, seed=21)
    d1 = rng.normal(size=(1500, 64)).astype(np.float32)
    d2 = d1 + 0.15 * rng.normal(size=d1.shape).astype(np.float32); d2[~lab] = rng.normal(size=((~lab).sum(), 64)).astype(np.float32)
    k1 = np.c_[p1, rng.uniform(4, 30, 1500), rng.uniform(0, 360, 1500)].astype(np.float32)
    k2 = np.c_[p2, rng.uniform(4, 30, 1500), rng.uniform(0, 360, 1500)].astype(np.float32)
    dev = torch.device("cuda", 0)
    q, t, d = tensor_api.match_snn_tensors(torch.from_numpy(d1).to(dev), torch.from_numpy(d2).to(dev), 0.9, mutual=True)
    hq, ht, hd = matcher.match_snn(d1, d2, 0.9, mutual=True)
    assert np.array_equal(q.cpu().numpy(), hq) and np.array_equal(t.cpu().numpy(), ht) and np.array_equal(d.cpu().numpy(), hd)
    A1 = tensor_api.kpts_to_xyA_tensors(torch.from_numpy(k1).to(dev)); A2 = tensor_api.kpts_to_xyA_tensors(torch.from_numpy(k2).to(dev))
    assert np.array_equal(A1.cpu().numpy(), matcher.kpts_to_xyA(k1))
    src = A1[q]; dst = A2[t]
    H, m, st, offs = tensor_api.find_homography_batch_tensors(src, dst, [int(q.numel())], 2.0, 0.999, 20000, -1.0, "sampson", True, seeds=[5])
    Hh, mh = pd.findHomography(src.cpu().numpy()[:, :2], dst.cpu().numpy()[:, :2], 2.0, 0.999, 20000, seed=5)
    assert np.array_equal(m.cpu().numpy(), np.asarray(mh)) and m.sum().item() > 0.8 * q.numel()
    assert np.linalg.norm(H[0].cpu().numpy() - Hh) <= 1e-9 * np.linalg.norm(Hh)
