"""TEST INFRASTRUCTURE ONLY.  ctypes loader for oracle/_ref/libdegensac_ref.so — the UNMODIFIED
reference C sources (compiled by oracle/Makefile from /root/reference) behind oracle/ref_shim.c.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import ctypes as C
import os
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def available(flavour=""):
    return os.path.exists(os.path.join(_HERE, "_ref", f"libdegensac_ref{flavour}.so"))


def lib(flavour=""):
    global _LIB
    if _LIB is None or getattr(_LIB, "_flavour", None) != flavour:
        os.environ.setdefault("OPENBLAS_NUM_THREADS", "1")
        os.environ.setdefault("MKL_NUM_THREADS", "1")
        l = C.CDLL(os.path.join(_HERE, "_ref", f"libdegensac_ref{flavour}.so"))
        dp = C.POINTER(C.c_double)
        l.ref_find_fundamental.argtypes = [dp, dp, C.c_int, C.c_int, C.c_double, C.c_double, C.c_int,
                                           C.c_int, C.c_int, C.c_double, C.c_int, C.c_uint, C.c_int,
                                           dp, C.POINTER(C.c_ubyte), C.POINTER(C.c_int)]
        l.ref_find_fundamental.restype = C.c_int
        l.ref_find_homography.argtypes = [dp, dp, C.c_int, C.c_int, C.c_double, C.c_double, C.c_int,
                                          C.c_int, C.c_int, C.c_double, C.c_uint, C.c_int,
                                          dp, C.POINTER(C.c_ubyte), C.POINTER(C.c_int)]
        l.ref_find_homography.restype = C.c_int
        l.ref_capture_data_out.argtypes = [C.POINTER(C.c_int), C.c_int]
        l.ref_capture_data_out.restype = None
        l.ref_capture_resids.argtypes = [dp, C.c_int]
        l.ref_capture_resids.restype = None
        l.ref_counters_reset.argtypes = [C.c_int]
        l.ref_counters_get.argtypes = [C.POINTER(C.c_longlong), C.POINTER(C.c_longlong), C.POINTER(C.c_double)]
        l.ref_time_to_best.argtypes = []
        l.ref_time_to_best.restype = C.c_double
        l.ref_find_fundamental_legacy.argtypes = [C.c_int, dp, dp, C.c_int, C.c_int, C.c_double, C.c_double, C.c_int, C.c_int, C.c_int, C.c_uint,
                                                  dp, C.POINTER(C.c_ubyte), C.POINTER(C.c_int)]
        l.ref_find_fundamental_legacy.restype = C.c_int
        l._flavour = flavour
        _LIB = l
    return _LIB


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def find_fundamental(pts1, pts2, px_th=0.5, conf=0.9999, max_iters=100000, error_type=0,
                     sym_check=True, laf_coef=0.0, degen=True, seed=1, count_models=False,
                     time_passes=False, flavour=""):
    """Reference exp_ransacFcustomLAF through the bindings.cpp-equivalent marshalling.
    Returns (F [3,3] row-major, mask [n] bool, stats dict)."""
    l = lib(flavour)
    a = np.ascontiguousarray(pts1, dtype=np.float64); b = np.ascontiguousarray(pts2, dtype=np.float64)
    n, dim = a.shape
    F = np.zeros(9); mask = np.zeros(n, np.uint8); st = (C.c_int * 4)()
    l.ref_counters_reset(int(time_passes))
    l.ref_find_fundamental(_dp(a), _dp(b), n, dim, px_th, conf, max_iters, error_type, int(sym_check),
                           max(0.0, laf_coef), int(degen), seed, int(count_models or time_passes),
                           _dp(F), mask.ctypes.data_as(C.POINTER(C.c_ubyte)), st)
    full = C.c_longlong(); ex = C.c_longlong(); sec = C.c_double()
    l.ref_counters_get(C.byref(full), C.byref(ex), C.byref(sec))
    stats = dict(samples=st[0], lo_runs=st[1], Ih=st[2], I=st[3], full_passes=full.value,
                 ex_passes=ex.value, models=full.value + ex.value, pass_seconds=sec.value,
                 time_to_best_s=l.ref_time_to_best())        # -1: not logged (count_models off) or the returned model was never scored
    return F.reshape(3, 3), mask.astype(bool), stats


def find_homography(pts1, pts2, px_th=1.0, conf=0.999, max_iters=50000, error_type=0,
                    sym_check=True, laf_coef=0.0, seed=1, count_models=False, time_passes=False,
                    flavour=""):
    """Reference exp_ransacHcustomLAF.  Returns the RAW internal H (column-wise, image2->image1;
    the Python wrapper applies inv(H.T), utils.py:108), mask, stats."""
    l = lib(flavour)
    a = np.ascontiguousarray(pts1, dtype=np.float64); b = np.ascontiguousarray(pts2, dtype=np.float64)
    n, dim = a.shape
    H = np.zeros(9); mask = np.zeros(n, np.uint8); st = (C.c_int * 4)()
    l.ref_counters_reset(int(time_passes))
    l.ref_find_homography(_dp(a), _dp(b), n, dim, px_th, conf, max_iters, error_type, int(sym_check),
                          max(0.0, laf_coef), seed, int(count_models or time_passes),
                          _dp(H), mask.ctypes.data_as(C.POINTER(C.c_ubyte)), st)
    full = C.c_longlong(); ex = C.c_longlong(); sec = C.c_double()
    l.ref_counters_get(C.byref(full), C.byref(ex), C.byref(sec))
    stats = dict(samples=st[0], lo_runs=st[1], rejected=st[2], I=st[3], full_passes=full.value,
                 models=full.value, pass_seconds=sec.value, time_to_best_s=l.ref_time_to_best())
    return H.reshape(3, 3), mask.astype(bool), stats


def resids_of(which, pts1, pts2, runs, **kw):
    """The residual dump the reference's driver fills per LO run and its binding frees unseen (RESIDS_M = 62 rows of n per
    run): the first `runs` LO runs as an array [runs, 62, n] (runs that did not happen stay NaN; rows the reference never
    writes inside a run hold whatever realloc returned), plus the usual result tuple."""
    n = np.asarray(pts1).shape[0]
    buf = np.full((runs, 62, n), np.nan)
    lib(kw.get("flavour", "")).ref_capture_resids(_dp(buf), runs)
    out = (find_fundamental if which == "F" else find_homography)(pts1, pts2, **kw)
    return buf, out


def data_out_of(pts1, pts2, **kw):
    """The F driver's `data_out` ([0] samples, [1] LO runs, [2 + I] samples whose best root had I inliers; allocated and freed
    unseen at bindings.cpp:412, :459): the first n + 3 ints, plus the usual result tuple."""
    n = np.asarray(pts1).shape[0]
    buf = np.zeros(n + 3, np.int32)
    lib(kw.get("flavour", "")).ref_capture_data_out(buf.ctypes.data_as(C.POINTER(C.c_int)), n + 3)
    out = find_fundamental(pts1, pts2, **kw)
    return buf, out


def find_fundamental_legacy(variant, pts1, pts2, px_th=0.5, conf=0.9999, max_iters=100000, error_type=0, sym_check=False, seed=1):
    """The reference's legacy drivers: variant 0 = exp_ransacFcustom (exp_ranF.c:811), 1 = exp_ransacF (:242, Sampson only).
    Returns (F [3,3], mask [n] bool, stats dict)."""
    l = lib()
    a = np.ascontiguousarray(pts1, dtype=np.float64); b = np.ascontiguousarray(pts2, dtype=np.float64)
    n, dim = a.shape
    F = np.zeros(9); mask = np.zeros(n, np.uint8); st = (C.c_int * 4)()
    l.ref_find_fundamental_legacy(int(variant), _dp(a), _dp(b), n, dim, px_th, conf, max_iters, error_type, int(sym_check), seed,
                                  _dp(F), mask.ctypes.data_as(C.POINTER(C.c_ubyte)), st)
    return F.reshape(3, 3), mask.astype(bool), dict(samples=st[0], lo_runs=st[1], Ih=st[2], I=st[3])


def ransacH2el(u10, th=4.0, conf=0.99, max_iters=10000, do_lo=True, inl_limit=0, seed=1):
    """The reference's ransacH2el (ranH2el.c:19), seeded through srand(seed).  Returns raw H, mask, stats."""
    l = lib()
    u = np.ascontiguousarray(u10, dtype=np.float64); n = u.shape[0]
    H = np.zeros(9); mask = np.zeros(n, np.uint8); st = (C.c_int * 4)()
    l.ref_ransacH2el(_dp(u), n, C.c_double(th), C.c_double(conf), int(max_iters), int(bool(do_lo)), int(inl_limit), C.c_uint(seed),
                     _dp(H), mask.ctypes.data_as(C.POINTER(C.c_ubyte)), st)
    return H.reshape(3, 3), mask.astype(bool), dict(samples=st[0], lo_runs=st[1], I=st[3])
