"""TEST INFRASTRUCTURE ONLY.  ctypes loader for oracle/libdg_oracle.so, the CPU restatement
(oracle/dg_oracle.c).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
import this module."""
import ctypes as C
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
ST_COUNT = 16


class Score(C.Structure):
    _fields_ = [("I", C.c_uint), ("J", C.c_double), ("Is", C.c_uint), ("Ilafs", C.c_uint)]


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE, "port"])


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "libdg_oracle.so")
        if not os.path.exists(path):
            build()
        l = C.CDLL(path)
        dp = C.POINTER(C.c_double); ip = C.POINTER(C.c_int)
        l.dg_oracle_find_fundamental.argtypes = [dp, dp, C.c_int, C.c_int, C.c_double, C.c_double, C.c_int, C.c_int,
                                                 C.c_int, C.c_double, C.c_int, C.c_uint, C.c_int, dp,
                                                 C.POINTER(C.c_ubyte), ip]
        l.dg_oracle_find_homography.argtypes = [dp, dp, C.c_int, C.c_int, C.c_double, C.c_double, C.c_int, C.c_int,
                                                C.c_int, C.c_double, C.c_uint, dp, C.POINTER(C.c_ubyte), ip]
        l.dg_oracle_inlidxs.restype = Score
        l.dg_oracle_inlidxs.argtypes = [dp, C.c_int, C.c_double, ip]
        l.dg_oracle_hash.restype = C.c_uint32
        _LIB = l
    return _LIB


def dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def ip(a):
    return a.ctypes.data_as(C.POINTER(C.c_int))


def find_fundamental(pts1, pts2, px_th=0.5, conf=0.9999, max_iters=100000, error_type=0, sym_check=True,
                     laf_coef=0.0, degen=True, seed=1, final_laf_filter=False, legacy=False):
    """legacy=True: the sample-budget rule of exp_ransacF / exp_ransacFcustom (exp_ranF.c:242, :811; laf_coef=0), incl. exp_ransacFcustom's own symmetric check when sym_check is set"""
    l = lib()
    a = np.ascontiguousarray(pts1, dtype=np.float64); b = np.ascontiguousarray(pts2, dtype=np.float64)
    n, dim = a.shape
    F = np.zeros(9); mask = np.zeros(n, np.uint8); st = np.zeros(ST_COUNT, np.int32)
    l.dg_oracle_find_fundamental(dp(a), dp(b), n, dim, px_th, conf, max_iters, error_type, int(sym_check),
                                 max(0.0, laf_coef), int(degen), seed, int(final_laf_filter) | (2 if legacy else 0), dp(F),
                                 mask.ctypes.data_as(C.POINTER(C.c_ubyte)), ip(st))
    stats = dict(samples=int(st[0]), lo_runs=int(st[1]), rejected=int(st[2]), I=int(st[3]), models=int(st[4]),
                 degen=int(st[5]), Ih=int(st[6]), best_sample=int(st[7]), full_passes=int(st[8]),
                 ex_passes=int(st[9]), hds_passes=int(st[10]), fds_direct=int(st[11]))
    return F.reshape(3, 3), mask.astype(bool), stats


def find_homography(pts1, pts2, px_th=1.0, conf=0.999, max_iters=50000, error_type=0, sym_check=True,
                    laf_coef=0.0, seed=1):
    l = lib()
    a = np.ascontiguousarray(pts1, dtype=np.float64); b = np.ascontiguousarray(pts2, dtype=np.float64)
    n, dim = a.shape
    H = np.zeros(9); mask = np.zeros(n, np.uint8); st = np.zeros(ST_COUNT, np.int32)
    l.dg_oracle_find_homography(dp(a), dp(b), n, dim, px_th, conf, max_iters, error_type, int(sym_check),
                                max(0.0, laf_coef), seed, dp(H), mask.ctypes.data_as(C.POINTER(C.c_ubyte)), ip(st))
    stats = dict(samples=int(st[0]), lo_runs=int(st[1]), rejected=int(st[2]), I=int(st[3]), models=int(st[4]),
                 best_sample=int(st[7]))
    return H.reshape(3, 3), mask.astype(bool), stats


def ransacH2el(u10, th=4.0, conf=0.99, max_iters=10000, do_lo=True, inl_limit=0, seed=1):
    """Restatement of ranH2el.c:19 (2 ellipse-to-ellipse correspondences per sample).  u10: [n, 10] = x1 y1 a1 b1 c1 x2 y2 a2 b2 c2.
    Returns the RAW internal H (column-wise, image 2 -> image 1), mask, stats."""
    l = lib()
    u = np.ascontiguousarray(u10, dtype=np.float64); n = u.shape[0]
    H = np.zeros(9); mask = np.zeros(n, np.uint8); st = np.zeros(ST_COUNT, np.int32)
    l.dg_oracle_ransacH2el(dp(u), n, C.c_double(th), C.c_double(conf), int(max_iters), int(bool(do_lo)), int(inl_limit), C.c_uint(seed),
                           dp(H), mask.ctypes.data_as(C.POINTER(C.c_ubyte)), ip(st))
    return H.reshape(3, 3), mask.astype(bool), dict(samples=int(st[0]), lo_runs=int(st[1]), I=int(st[3]), models=int(st[4]))
