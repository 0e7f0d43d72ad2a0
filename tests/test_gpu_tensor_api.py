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


def test_fundamental_tensors_match_host_batch():
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
    assert (st.cpu().numpy()[:, 0] > 0).all()                          # samples drawn


def test_homography_tensors_match_host_batch():
    import torch
    sizes = [900, 300, 1500]
    A, B = _pairs("H", sizes); seeds = [2, 4, 6]
    Hh, mh = pd.findHomographyBatch(A, B, 1.0, 0.999, 20000, 3.0, "symm_max", True, seeds=seeds)
    dev = torch.device("cuda", 0)
    H, m, st, offs = tensor_api.find_homography_batch_tensors(torch.from_numpy(np.concatenate(A)).to(dev),
                                                              torch.from_numpy(np.concatenate(B)).to(dev), sizes, 1.0, 0.999, 20000,
                                                              3.0, "symm_max", True, seeds=seeds)
    H = H.cpu().numpy(); m = m.cpu().numpy()
    for p in range(3):
        assert np.array_equal(m[offs[p]:offs[p + 1]], np.asarray(mh[p]))
        # inv() runs in numpy on the host path and in torch.linalg on the device path: same to rounding
        assert np.linalg.norm(H[p] - Hh[p]) <= 1e-9 * max(np.linalg.norm(Hh[p]), 1e-300)
