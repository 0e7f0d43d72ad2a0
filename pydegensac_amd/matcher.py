"""Tentative correspondences on the GPU: the matcher stage of the reference's example pipeline
(examples/simple-example.py:46-53) behind libmi_degensac.so (include/mi_degensac.h, mi_degensac_match*).

    bf = cv2.BFMatcher(); matches = bf.knnMatch(descs1, descs2, k=2)          ->  idx, dist = knn_match(descs1, descs2)
    tentatives = [m for m, n in matches if m.distance < 0.9 * n.distance]     ->  q, t, d = match_snn(descs1, descs2, 0.9)

float32 descriptors use the L2 norm (cv2.BFMatcher's default, what the example runs on AKAZE's KAZE descriptors), uint8
descriptors the Hamming norm.  No CPU path: without the HIP library / a gfx950 device every call raises."""
import ctypes as C

import numpy as np

from . import _lib

NORM_L2, NORM_HAMMING = 0, 1


def _prep(desc1, desc2, norm):
    a = np.asarray(desc1); b = np.asarray(desc2)
    if a.ndim != 2 or b.ndim != 2 or a.shape[1] != b.shape[1] or a.shape[1] == 0:
        raise ValueError("descriptors should be arrays [n1, dim] and [n2, dim] with the same dim")
    if norm is None:
        norm = "hamming" if a.dtype == np.uint8 and b.dtype == np.uint8 else "l2"
    if norm not in ("l2", "hamming"):
        raise ValueError("norm should be 'l2' or 'hamming'")
    if norm == "l2":
        return NORM_L2, np.ascontiguousarray(a, np.float32), np.ascontiguousarray(b, np.float32)
    if a.dtype != np.uint8 or b.dtype != np.uint8:
        raise ValueError("the Hamming norm needs uint8 descriptors")
    pad = (-a.shape[1]) % 4                                   # whole 32-bit words; zero bytes add no differing bits
    if pad:
        a = np.pad(a, ((0, 0), (0, pad))); b = np.pad(b, ((0, 0), (0, pad)))
    return NORM_HAMMING, np.ascontiguousarray(a), np.ascontiguousarray(b)


def _run(desc1, desc2, norm, ratio, mutual, want_keep, device):
    code, a, b = _prep(desc1, desc2, norm)
    n1, n2, dim = a.shape[0], b.shape[0], a.shape[1]
    idx = np.full((n1, 2), -1, np.int32); dist = np.full((n1, 2), np.inf, np.float32)
    keep = np.zeros(n1, np.uint8) if want_keep else None
    rc = _lib.lib().mi_degensac_match(code, a.ctypes.data_as(C.c_void_p), n1, b.ctypes.data_as(C.c_void_p), n2, dim, float(ratio),
                                      int(bool(mutual)), int(device), idx.ctypes.data_as(C.POINTER(C.c_int32)),
                                      dist.ctypes.data_as(C.POINTER(C.c_float)),
                                      keep.ctypes.data_as(C.POINTER(C.c_uint8)) if want_keep else None)
    if rc != 0:
        msg = _lib.lib().mi_degensac_match_last_error().decode()
        if rc == -1:
            raise ValueError(msg)
        raise _lib.MiDegensacError(f"mi_degensac error {rc}: {msg}")
    return idx, dist, keep


def knn_match(desc1, desc2, norm=None, device=0):
    """The two nearest rows of desc2 for every row of desc1 (cv2 `knnMatch(descs1, descs2, k=2)`): idx [n1, 2] (train
    indices, -1 where desc2 has fewer than two rows) and dist [n1, 2], nearest first, ties to the lower index."""
    idx, dist, _ = _run(desc1, desc2, norm, 1.0, False, False, device)
    return idx, dist


def match_snn(desc1, desc2, ratio=0.9, mutual=False, norm=None, device=0):
    """Second-nearest-neighbour ratio test (`m.distance < ratio * n.distance`), optionally restricted to mutual nearest
    neighbours: (query indices, train indices, distances) of the tentative correspondences, in query order."""
    idx, dist, keep = _run(desc1, desc2, norm, ratio, mutual, True, device)
    sel = np.flatnonzero(keep)
    return sel.astype(np.int64), idx[sel, 0].astype(np.int64), dist[sel, 0]


def tentative_points(kps1, kps2, desc1, desc2, ratio=0.9, mutual=False, norm=None, device=0):
    """Matched coordinates ready for findHomography / findFundamentalMatrix: kps are [n, >=2] arrays (x, y, ...)"""
    q, t, _ = match_snn(desc1, desc2, ratio, mutual, norm, device)
    a = np.asarray(kps1, np.float64); b = np.asarray(kps2, np.float64)
    return a[q], b[t]


def kpts_to_xyA(kpts, device=0):
    """utils.py:24-41 `convert_cv2_kpts_to_xyA` on the GPU for keypoints given as an array [n, 4] = (x, y, size, angle in
    degrees) (cv2.KeyPoint.pt / .size / .angle): the [n, 6] float64 rows (x, y, a11, a12, a21, a22) the estimators accept."""
    k = np.ascontiguousarray(kpts, np.float32)
    if k.ndim != 2 or k.shape[1] != 4:
        raise ValueError("keypoints should be an array [n, 4] = (x, y, size, angle)")
    out = np.zeros((k.shape[0], 6))
    rc = _lib.lib().mi_degensac_kpts_to_xyA(k.ctypes.data_as(C.POINTER(C.c_float)), k.shape[0], int(device), _lib.dptr(out))
    if rc != 0:
        raise _lib.MiDegensacError(f"mi_degensac error {rc}: {_lib.lib().mi_degensac_match_last_error().decode()}")
    return out
