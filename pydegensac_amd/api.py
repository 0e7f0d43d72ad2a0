"""Host-side mirror of the reference's Python layer (src/pydegensac/utils.py) over the C-ABI.

Argument names, positional order, defaults, error-type strings, exception classes and return
types follow utils.py:74-146; the two private functions mirror the pybind signatures
(bindings.cpp:484-503).  Differences, all additive:
  * `seed` (default: time-based, like the reference's srand(time(NULL))) and `device` keywords;
  * inputs are forced C-contiguous (the reference silently mis-reads F-ordered arrays);
  * a model that was never found is returned as zeros (the reference returns uninitialised stack
    memory, bindings.cpp:110,321) so that the "no model -> all-False list" rule is deterministic.
"""
import ctypes as C
import math
import time
import warnings

import numpy as np

from . import _lib

try:  # utils.py:7-11
    import cv2
    OPENCV_HERE = True
except Exception:  # pragma: no cover
    OPENCV_HERE = False

error_type_dict_homography = {"sampson": 0, "symm_sq_max": 1, "symm_max": 2, "symm_sq_sum": 3, "symm_sum": 4}
error_type_dict_fundamental = {"sampson": 0, "symm_epipolar": 1}

import threading

_tls = threading.local()


def last_stats():
    """Statistics of the calling thread's most recent call (dict, or list of dicts for a batch)."""
    return getattr(_tls, "stats", None)


def convert_cv2_kpts_to_xyA(kps):
    """cv2.KeyPoint list -> [N,6] (x, y, a11, a12, a21, a22)   (utils.py:24-41)"""
    num = len(kps)
    out = np.zeros((num, 6)).astype(np.float64)
    for i, kp in enumerate(kps):
        out[i, :2] = kp.pt
        s = kp.size
        a = kp.angle
        cos = math.cos(a * math.pi / 180.0)
        sin = math.sin(a * math.pi / 180.0)
        out[i, 2] = s * cos
        out[i, 3] = s * sin
        out[i, 4] = -s * sin
        out[i, 5] = s * cos
    return out


def convert_and_check(kps1):
    """utils.py:43-71"""
    if type(kps1) is np.ndarray:
        sh = kps1.shape
        err_message = ValueError("Keypoints should be list of cv2.KeyPoint \
                             or numpy.array [Nx2] or [Nx6]. N>=4  \
                             Got shape of {} with shape instead".format(str(sh)))
        if len(sh) != 2:
            raise err_message
        num, dim = sh
        if (dim != 2) and (dim != 6):
            raise err_message
        if num < 4:
            raise err_message
        out = np.ascontiguousarray(kps1.astype(np.float64))
    elif type(kps1) is list:
        if OPENCV_HERE:
            if type(kps1[0]) is not cv2.KeyPoint:
                raise ValueError("Keypoints should be list of cv2.KeyPoint \
                                or numpy.array [Nx2] or [Nx6]. N>=4  \
                                Got input of list of type {}".format(str(type(kps1[0]))))
            else:
                out = convert_cv2_kpts_to_xyA(kps1)
        else:
            raise ValueError("Cannot import cv2. Please, install or pass np.arrays instead")
    else:
        raise ValueError("Keypoints should be list of cv2.KeyPoint \
                             or numpy.array [Nx2] or [Nx6]. N>=4  \
                             Got input of type {}".format(str(type(kps1))))
    return out


def _time_seed():
    return int(time.time()) & 0xFFFFFFFF


def _call_single(which, x1y1, x2y2, px_th, conf, max_iters, error_type, sym_check_enable, laf_coef, degen, seed, device,
                 flags=0, tuning=0):
    a = np.ascontiguousarray(x1y1, dtype=np.float64)
    b = np.ascontiguousarray(x2y2, dtype=np.float64)
    if a.ndim != 2 or b.ndim != 2:
        raise ValueError("x1y1 should be an array with dims [n,2] or [n,6]")
    n, dim = a.shape
    if b.shape[0] != n:
        raise ValueError("x1y1 and x2y2 should be the same size")          # bindings.cpp:45-47, :280-282
    if b.shape[1] != dim:
        raise ValueError("x1y1 and x2y2 should have the same number of columns")
    if dim not in (2, 6):
        raise ValueError("x1y1 should be an array with dims [n,2] or [n,6]")   # bindings.cpp:32-41, :267-276
    prm = _lib.make_params(px_th, conf, max_iters, error_type, sym_check_enable, laf_coef, degen, flags, tuning)
    model = np.zeros(9, np.float64)
    mask = np.zeros(n, np.uint8)
    st = np.zeros(_lib.STATS_LEN, np.int32)
    fn = _lib.lib().mi_degensac_find_fundamental if which == "F" else _lib.lib().mi_degensac_find_homography
    rc = fn(_lib.dptr(a), _lib.dptr(b), n, dim, C.byref(prm), int(seed) & 0xFFFFFFFF, int(device), _lib.dptr(model),
            mask.ctypes.data_as(C.POINTER(C.c_uint8)), st.ctypes.data_as(C.POINTER(C.c_int32)))
    _lib.check(rc)
    _tls.stats = _lib.stats_dict(st)
    return model.reshape(3, 3), mask.astype(bool)


def findHomography_(x1y1, x2y2, px_th=1.0, conf=0.999, max_iters=10000, error_type=0, sym_check_enable=True,
                    laf_coef=0.0, seed=None, device=0, tuning=0):
    """bindings.cpp:19-251; defaults of the pybind definition, bindings.cpp:484-492.
    Returns the driver's raw H (column-wise, image2->image1) and the mask."""
    return _call_single("H", x1y1, x2y2, px_th, conf, max_iters, error_type, sym_check_enable, laf_coef, True,
                        _time_seed() if seed is None else seed, device, 0, tuning)


def findFundamentalMatrix_(x1y1, x2y2, px_th=0.5, conf=0.9999, max_iters=200000, error_type=0, sym_check_enable=True,
                           laf_coef=0.0, enable_degeneracy_check=True, seed=None, device=0, flags=0, tuning=0):
    """bindings.cpp:253-467; defaults of the pybind definition, bindings.cpp:494-503."""
    return _call_single("F", x1y1, x2y2, px_th, conf, max_iters, error_type, sym_check_enable, laf_coef,
                        enable_degeneracy_check, _time_seed() if seed is None else seed, device, flags, tuning)


def findHomography(pts1_, pts2_, px_th=1.0, conf=0.999, max_iters=50000, laf_consistensy_coef=-1.0,
                   error_type="sampson", symmetric_error_check=True, seed=None, device=0):
    """utils.py:74-109"""
    pts1 = convert_and_check(pts1_)
    pts2 = convert_and_check(pts2_)
    n, dim = pts1.shape
    n2, dim2 = pts2.shape
    assert (n == n2) and (dim == dim2)
    if dim == 2 and laf_consistensy_coef > 0:
        warnings.warn('You set laf_consistensy_coef, but provided only (x,y) keypoints. Skipping LAF check')
        laf_consistensy_coef = 0
    try:
        error_type_int = error_type_dict_homography[error_type.lower()]
    except Exception:
        raise ValueError("Error type should be on of {}. Got {} instead".format(list(error_type_dict_homography.keys()),
                                                                           error_type))
    laf_consistensy_coef = max(0, laf_consistensy_coef)
    H, mask = findHomography_(pts1, pts2, px_th, conf, max_iters, error_type_int, symmetric_error_check,
                              laf_consistensy_coef, seed=seed, device=device)
    if np.abs(H).sum() == 0:
        mask = [False] * len(mask)
        return H, mask
    H_out = np.linalg.inv(H.T)
    return H_out, mask


def findFundamentalMatrix(pts1_, pts2_, px_th=0.5, conf=0.9999, max_iters=100000, laf_consistensy_coef=-1.0,
                          error_type="sampson", symmetric_error_check=True, enable_degeneracy_check=True,
                          seed=None, device=0):
    """utils.py:111-146"""
    pts1 = convert_and_check(pts1_)
    pts2 = convert_and_check(pts2_)
    n, dim = pts1.shape
    n2, dim2 = pts2.shape
    assert (n == n2) and (dim == dim2)
    if dim == 2 and laf_consistensy_coef > 0:
        warnings.warn('You set laf_consistensy_coef, but provided only (x,y) keypoints. Skipping LAF check')
        laf_consistensy_coef = 0
    try:
        error_type_int = error_type_dict_fundamental[error_type.lower()]
    except Exception:
        raise ValueError("Error type should be on of {}. Got {} instead".format(list(error_type_dict_fundamental.keys()),
                                                                           error_type))
    laf_consistensy_coef = max(0, laf_consistensy_coef)
    if n < 8:
        raise ValueError("x1y1 should be an array with dims [n,2], n>=8")     # bindings.cpp:270-272
    F, mask = findFundamentalMatrix_(pts1, pts2, px_th, conf, max_iters, error_type_int, symmetric_error_check,
                                     laf_consistensy_coef, enable_degeneracy_check, seed=seed, device=device)
    if np.abs(F).sum() == 0:
        mask = [False] * n
    return F, mask


def _error_type(table, error_type):
    try:
        return table[error_type.lower()]
    except Exception:
        raise ValueError("Error type should be on of {}. Got {} instead".format(list(table.keys()), error_type))


def _batch(which, pts1_list, pts2_list, px_th, conf, max_iters, error_type_int, sym, laf, degen, seeds, device, tuning=0, flags=0, devices=None):
    n_pairs = len(pts1_list)
    if n_pairs == 0 or len(pts2_list) != n_pairs:
        raise ValueError("pts1_list and pts2_list must hold the same, non-zero number of pairs")
    a = [np.ascontiguousarray(p, dtype=np.float64) for p in pts1_list]
    b = [np.ascontiguousarray(p, dtype=np.float64) for p in pts2_list]
    if a[0].ndim != 2 or a[0].shape[1] not in (2, 6):
        raise ValueError("x1y1 should be an array with dims [n,2] or [n,6]")
    dim = a[0].shape[1]
    for i in range(n_pairs):
        # same checks as the single-pair path (bindings.cpp:32-47): per pair, both sides [n_i, dim]
        if a[i].ndim != 2 or a[i].shape[1] != dim:
            raise ValueError(f"pair {i}: x1y1 should be an array with dims [n,{dim}] like pair 0")
        if b[i].shape != a[i].shape:
            raise ValueError(f"pair {i}: x1y1 and x2y2 should be the same size")
    if len(seeds) != n_pairs:
        raise ValueError("one seed per pair")
    offs = np.zeros(n_pairs + 1, np.int64)
    offs[1:] = np.cumsum([x.shape[0] for x in a])
    A = np.ascontiguousarray(np.concatenate(a, 0)); B = np.ascontiguousarray(np.concatenate(b, 0))
    prm = _lib.make_params(px_th, conf, max_iters, error_type_int, sym, laf, degen, flags, tuning)
    model = np.zeros((n_pairs, 9)); mask = np.zeros(int(offs[-1]), np.uint8); st = np.zeros((n_pairs, 16), np.int32)
    sd = np.ascontiguousarray(seeds, dtype=np.uint32)
    if devices is not None:
        # one process, several GPUs: device devices[k] gets the k-th contiguous block of pairs, one host thread each (no collective)
        dv = np.ascontiguousarray(list(devices), dtype=np.int32)
        if dv.ndim != 1 or dv.size == 0:
            raise ValueError("devices must be a non-empty list of device indices")
        fn = _lib.lib().mi_degensac_find_fundamental_batch_multi if which == "F" else _lib.lib().mi_degensac_find_homography_batch_multi
        rc = fn(_lib.dptr(A), _lib.dptr(B), offs.ctypes.data_as(C.POINTER(C.c_int64)), n_pairs, dim, C.byref(prm),
                sd.ctypes.data_as(C.POINTER(C.c_uint32)), dv.ctypes.data_as(C.POINTER(C.c_int32)), int(dv.size), _lib.dptr(model),
                mask.ctypes.data_as(C.POINTER(C.c_uint8)), st.ctypes.data_as(C.POINTER(C.c_int32)))
    else:
        fn = _lib.lib().mi_degensac_find_fundamental_batch if which == "F" else _lib.lib().mi_degensac_find_homography_batch
        rc = fn(_lib.dptr(A), _lib.dptr(B), offs.ctypes.data_as(C.POINTER(C.c_int64)), n_pairs, dim, C.byref(prm),
                sd.ctypes.data_as(C.POINTER(C.c_uint32)), int(device), _lib.dptr(model),
                mask.ctypes.data_as(C.POINTER(C.c_uint8)), st.ctypes.data_as(C.POINTER(C.c_int32)))
    _lib.check(rc)
    _tls.stats = [_lib.stats_dict(s) for s in st]
    masks = [mask[offs[i]:offs[i + 1]].astype(bool) for i in range(n_pairs)]
    return model.reshape(n_pairs, 3, 3), masks


def findFundamentalMatrixBatch(pts1_list, pts2_list, px_th=0.5, conf=0.9999, max_iters=100000,
                               laf_consistensy_coef=-1.0, error_type="sampson", symmetric_error_check=True,
                               enable_degeneracy_check=True, seeds=None, device=0, tuning=0, flags=0, devices=None):
    """Independent image pairs in one launch (persistent workgroups, one pair at a time each).
    devices=[d0, d1, ...]: ONE process drives several GPUs — device dk gets the k-th contiguous block of pairs (as
    parallel.shard_range), one host thread per device, results gathered on the host; seeds travel with their pairs, so the
    results are those of one device.  Returns (F [P,3,3], [mask_p])."""
    et = _error_type(error_type_dict_fundamental, error_type)
    if seeds is None:
        seeds = (_time_seed() + np.arange(len(pts1_list))) & 0xFFFFFFFF
    return _batch("F", pts1_list, pts2_list, px_th, conf, max_iters, et, symmetric_error_check,
                  max(0, laf_consistensy_coef), enable_degeneracy_check, seeds, device, tuning, flags, devices)


def ransacF_legacy(pts1, pts2, px_th=0.5, conf=0.9999, max_iters=100000, error_type="sampson", seed=None, device=0, tuning=0,
                   symmetric_error_check=False):
    """The reference's older fundamental-matrix drivers, which its Python module no longer reaches: `exp_ransacF`
    (exp_ranF.c:242; error_type "sampson") and `exp_ransacFcustom` (exp_ranF.c:811; either metric; symmetric_error_check =
    its own symmetric check: all points, 16 px_th^2, the final mask filtered with the model the driver computed last).  They
    differ from findFundamentalMatrix in one rule: the sample budget follows every new best model, also one found between
    two local optimisations (MI_DEGENSAC_FLAG_LEGACY_F).  Returns (F [3,3], mask [n] bool)."""
    et = _error_type(error_type_dict_fundamental, error_type)
    return _call_single("F", convert_and_check(pts1), convert_and_check(pts2), px_th, conf, max_iters, et, bool(symmetric_error_check), 0.0, True,
                        _time_seed() if seed is None else seed, device, _lib.FLAG_LEGACY_F, tuning)


def ransacF_legacy_batch(pts1_list, pts2_list, px_th=0.5, conf=0.9999, max_iters=100000, error_type="sampson", seeds=None,
                         device=0, tuning=0, symmetric_error_check=False):
    """ransacF_legacy for independent pairs in one launch.  Returns (F [P,3,3], [mask_p])."""
    et = _error_type(error_type_dict_fundamental, error_type)
    if seeds is None:
        seeds = (_time_seed() + np.arange(len(pts1_list))) & 0xFFFFFFFF
    return _batch("F", pts1_list, pts2_list, px_th, conf, max_iters, et, bool(symmetric_error_check), 0.0, True, seeds, device, tuning, _lib.FLAG_LEGACY_F)


def findHomographyBatch(pts1_list, pts2_list, px_th=1.0, conf=0.999, max_iters=50000, laf_consistensy_coef=-1.0,
                        error_type="sampson", symmetric_error_check=True, seeds=None, device=0, tuning=0, flags=0, devices=None):
    """Batch homographies; returns the user-facing H_out = inv(H.T) per pair (zeros when none found).
    devices=[...]: one process, several GPUs (see findFundamentalMatrixBatch)."""
    et = _error_type(error_type_dict_homography, error_type)
    if seeds is None:
        seeds = (_time_seed() + np.arange(len(pts1_list))) & 0xFFFFFFFF
    H, masks = _batch("H", pts1_list, pts2_list, px_th, conf, max_iters, et, symmetric_error_check,
                      max(0, laf_consistensy_coef), True, seeds, device, tuning, flags, devices)
    out = np.zeros_like(H)
    for i in range(len(H)):
        if np.abs(H[i]).sum() != 0:
            out[i] = np.linalg.inv(H[i].T)
    return out, masks


def ransacH2el_batch(u10_list, th=4.0, conf=0.99, max_iters=10000, do_lo=True, inl_limit=0, seeds=None, device=0, raw=False):
    """The reference's ransacH2el (degensac/ranH2el.c:19, ranH2el.h:35; it has no Python binding there): RANSAC on
    ellipse-to-ellipse correspondences, two per sample.  u10: [n, 10] = x1 y1 a1 b1 c1 x2 y2 a2 b2 c2 with the local affine
    frame [a 0; b c] of each image.  th is the threshold on the squared transfer error.  Returns (H [P, 3, 3], list of
    masks); `last_stats()` has the counters.  H follows findHomography's convention: the conventional row-major matrix that
    maps image 1 to image 2, H_out = inv(H_raw.T) (utils.py:108), zeros — with an all-false mask, as findHomography does
    (utils.py:104-107) — when no model was found or the raw model is singular.  (Before round 4 this function returned the raw
    array; callers that relied on that pass raw=True.)  raw=True returns the driver's own array instead: the reference's internal model as the C-ABI documents it,
    9 doubles stored column-wise that map image 2 to image 1 — reshaped [3, 3] it is the TRANSPOSE of that matrix."""
    n_pairs = len(u10_list)
    if n_pairs == 0:
        raise ValueError("u10_list must hold at least one pair")
    a = [np.ascontiguousarray(p, dtype=np.float64) for p in u10_list]
    for i, x in enumerate(a):
        if x.ndim != 2 or x.shape[1] != 10 or x.shape[0] < 2:
            raise ValueError(f"pair {i}: u10 should be an array with dims [n,10], n >= 2")
    if seeds is None:
        seeds = (_time_seed() + np.arange(n_pairs)) & 0xFFFFFFFF
    if len(seeds) != n_pairs:
        raise ValueError("one seed per pair")
    offs = np.zeros(n_pairs + 1, np.int64)
    offs[1:] = np.cumsum([x.shape[0] for x in a])
    U = np.ascontiguousarray(np.concatenate(a, 0))
    prm = _lib.H2elParams(float(th), float(conf), int(max_iters), int(bool(do_lo)), int(inl_limit), 0)
    model = np.zeros((n_pairs, 9)); mask = np.zeros(int(offs[-1]), np.uint8); st = np.zeros((n_pairs, 16), np.int32)
    sd = np.ascontiguousarray(seeds, dtype=np.uint32)
    rc = _lib.lib().mi_degensac_ransac_h2el_batch(_lib.dptr(U), offs.ctypes.data_as(C.POINTER(C.c_int64)), n_pairs, C.byref(prm),
                                                  sd.ctypes.data_as(C.POINTER(C.c_uint32)), int(device), _lib.dptr(model),
                                                  mask.ctypes.data_as(C.POINTER(C.c_uint8)), st.ctypes.data_as(C.POINTER(C.c_int32)))
    _lib.check(rc)
    _tls.stats = [_lib.stats_dict(s) for s in st]
    masks = [mask[offs[i]:offs[i + 1]].astype(bool) for i in range(n_pairs)]
    Hr = model.reshape(n_pairs, 3, 3)
    if raw:
        return Hr, masks
    out = np.zeros_like(Hr)
    for i in range(n_pairs):
        if np.abs(Hr[i]).sum() != 0:
            try:
                out[i] = np.linalg.inv(Hr[i].T)
            except np.linalg.LinAlgError:
                pass
        if np.abs(out[i]).sum() == 0:
            masks[i] = np.zeros_like(masks[i])       # findHomography's convention: no model, no inliers (utils.py:104-107)
    return out, masks


def ransacH2el(u10, th=4.0, conf=0.99, max_iters=10000, do_lo=True, inl_limit=0, seed=None, device=0, raw=False):
    """One pair of ransacH2el_batch: returns (H [3, 3], mask [n])."""
    H, m = ransacH2el_batch([u10], th, conf, max_iters, do_lo, inl_limit, None if seed is None else [seed], device, raw)
    _tls.stats = _tls.stats[0]
    return H[0], m[0]
