"""Structured diagnostics: what the reference's drivers compute and its binding discards
(bindings.cpp:242-243, :458-459).  `*_with_residuals` run one pair like findFundamentalMatrix_ / findHomography_ and also
return the per-LO residual dump (include/mi_degensac.h, MI_DEGENSAC_RESIDS_M): an array [runs, 62, n] whose rows are

    0        residuals of the so-far-the-best sample's model when the LO started
    1        residuals of the least-squares model fitted before the LO
    2 + 6 i  repetition i of the LO: the model of its random inlier subset
    3 + 6 i .. 6 + 6 i   its four inner iterations (the LO's own metric)
    7 + 6 i  its final pass (the driver's metric)

NaN rows were never computed (early exits); LO runs that did not happen are all NaN."""
import ctypes as C

import numpy as np

from . import _lib

RESIDS_M = 62


def _run(which, x1y1, x2y2, px_th, conf, max_iters, error_type, sym, laf_coef, degen, seed, device, runs):
    a = np.ascontiguousarray(x1y1, np.float64); b = np.ascontiguousarray(x2y2, np.float64)
    if a.ndim != 2 or a.shape != b.shape or a.shape[1] not in (2, 6):
        raise ValueError("x1y1 and x2y2 should be arrays of the same shape [n,2] or [n,6]")
    n, dim = a.shape
    prm = _lib.make_params(px_th, conf, max_iters, error_type, sym, laf_coef, degen)
    model = np.zeros(9); mask = np.zeros(n, np.uint8); st = np.zeros(_lib.STATS_LEN, np.int32)
    res = np.empty((runs, RESIDS_M, n))
    fn = _lib.lib().mi_degensac_find_fundamental_resids if which == "F" else _lib.lib().mi_degensac_find_homography_resids
    _lib.check(fn(_lib.dptr(a), _lib.dptr(b), n, dim, C.byref(prm), int(seed) & 0xFFFFFFFF, int(device), _lib.dptr(model),
                  mask.ctypes.data_as(C.POINTER(C.c_uint8)), st.ctypes.data_as(C.POINTER(C.c_int32)), _lib.dptr(res), int(runs)))
    return model.reshape(3, 3), mask.astype(bool), _lib.stats_dict(st), res


def find_fundamental_with_residuals(x1y1, x2y2, px_th=0.5, conf=0.9999, max_iters=200000, error_type=0, sym_check_enable=True,
                                    laf_coef=0.0, enable_degeneracy_check=True, seed=1, device=0, lo_runs=16):
    """(F, mask, stats, resids [lo_runs, 62, n]) — F as findFundamentalMatrix_ returns it"""
    return _run("F", x1y1, x2y2, px_th, conf, max_iters, error_type, sym_check_enable, laf_coef, enable_degeneracy_check, seed, device, lo_runs)


def find_homography_with_residuals(x1y1, x2y2, px_th=1.0, conf=0.999, max_iters=10000, error_type=0, sym_check_enable=True,
                                   laf_coef=0.0, seed=1, device=0, lo_runs=16):
    """(H, mask, stats, resids [lo_runs, 62, n]) — H as findHomography_ returns it (the driver's raw model)"""
    return _run("H", x1y1, x2y2, px_th, conf, max_iters, error_type, sym_check_enable, laf_coef, True, seed, device, lo_runs)


def find_fundamental_with_support_histogram(x1y1, x2y2, px_th=0.5, conf=0.9999, max_iters=200000, error_type=0, sym_check_enable=True,
                                            laf_coef=0.0, enable_degeneracy_check=True, seed=1, device=0):
    """(F, mask, stats, hist) — hist = the reference driver's `data_out` (exp_ranF.c:1495, :1758-1759): hist[0] samples drawn,
    hist[1] LO runs, hist[2 + I] = number of samples whose best model had exactly I inliers.  Every model is scored exactly
    for this (no screening): same result, slower call."""
    a = np.ascontiguousarray(x1y1, np.float64); b = np.ascontiguousarray(x2y2, np.float64)
    if a.ndim != 2 or a.shape != b.shape or a.shape[1] not in (2, 6):
        raise ValueError("x1y1 and x2y2 should be arrays of the same shape [n,2] or [n,6]")
    n, dim = a.shape
    prm = _lib.make_params(px_th, conf, max_iters, error_type, sym_check_enable, laf_coef, enable_degeneracy_check)
    model = np.zeros(9); mask = np.zeros(n, np.uint8); st = np.zeros(_lib.STATS_LEN, np.int32); hist = np.zeros(n + 3, np.int32)
    _lib.check(_lib.lib().mi_degensac_find_fundamental_hist(_lib.dptr(a), _lib.dptr(b), n, dim, C.byref(prm), int(seed) & 0xFFFFFFFF, int(device),
                                                            _lib.dptr(model), mask.ctypes.data_as(C.POINTER(C.c_uint8)),
                                                            st.ctypes.data_as(C.POINTER(C.c_int32)), hist.ctypes.data_as(C.POINTER(C.c_int32))))
    return model.reshape(3, 3), mask.astype(bool), _lib.stats_dict(st), hist
