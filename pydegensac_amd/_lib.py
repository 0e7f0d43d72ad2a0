"""ctypes binding of libmi_degensac.so (include/mi_degensac.h).  No CPU fallback: if the HIP
library is missing or no gfx950 device is usable, every call raises."""
import ctypes as C
import os
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MI_DEGENSAC_LIB") or os.path.join(_HERE, "libmi_degensac.so")
STATS_LEN = 16
STAT_NAMES = ["samples", "lo_runs", "rejected", "I", "models", "degen", "Ih", "best_sample",
              "full_passes", "ex_passes", "h_passes", "aux_passes", "ticks_best", "ticks_total", "threads", "placement"]
FLAG_FINAL_LAF_FILTER = 1
FLAG_LEGACY_F = 2            # exp_ransacF / exp_ransacFcustom sample-budget rule (include/mi_degensac.h)
# per-call scheduling switches (results never depend on them; they win over set_stream_mode / set_hjob_mode)
FLAG_NO_STREAM, FLAG_STREAM_ON, FLAG_NO_HJOB, FLAG_STREAM_AUTO, FLAG_HJOB_ON = 4, 8, 16, 64, 128


def FLAG_STREAM_TEST(b): return (int(b) & 3) << 8        # noqa: E704  with FLAG_STREAM_ON: bit 0 = owner re-scores, bit 1 = ask at once
# params.tuning (include/mi_degensac.h MI_DEGENSAC_TUNE_*): speed knobs only, results never depend on them
TUNE_LATENCY, TUNE_THROUGHPUT, TUNE_THROUGHPUT4 = 1, 2, 3   # kernel variant: 512- / 256- / 128-thread workgroups
TUNE_PLACE_HBM, TUNE_PLACE_LDS, TUNE_PLACE_POOL_LDS = 1 << 2, 2 << 2, 3 << 2
TUNE_SEQ_POOL = 1 << 4
TUNE_H_SERIAL_LO = 1 << 5   # homography only: local-optimisation repetitions one after the other (default: one per wave)
TUNE_COOP_ALL_PASSES = 1 << 6   # fundamental matrix with helper workgroups: distribute every full pass (tests)
TUNE_F_SERIAL_REPS = 1 << 7   # fundamental matrix: the repetitions of innerH and of the local optimisation one after the other (default: one per wave; tests)


def TUNE_HELPERS(h): return (int(h) & 255) << 8          # noqa: E704  helper workgroups per pair (255 = off)
def TUNE_SET_ASIDE(t): return (int(t) & 255) << 16       # noqa: E704  set pairs aside after t * 256 samples (255 = off)
def TUNE_GRID_CAP(g): return (int(g) & 31) << 24         # noqa: E704  cap on resident workgroups (tests)
def TUNE_LONG_SHIFT(l): return (int(l) & 7) << 29        # noqa: E704,E741  "many samples left" = threshold << l
# bits 8-15: cooperative helper workgroups per pair (0 auto, 255 off); bits 16-23: samples after which a running pair is
# set aside while unstarted pairs remain, in units of 256 (0 auto, 255 off); bits 24-31: cap on resident workgroups (tests)


class Params(C.Structure):
    _fields_ = [("px_th", C.c_double), ("conf", C.c_double), ("max_iters", C.c_int32), ("error_type", C.c_int32),
                ("symmetric_error_check", C.c_int32), ("enable_degeneracy_check", C.c_int32),
                ("laf_consistensy_coef", C.c_double), ("flags", C.c_uint32), ("tuning", C.c_uint32)]


class Diag(C.Structure):
    """mi_degensac_diag (include/mi_degensac.h): optional device buffers of the *_batch_dev_ex entry points.  struct_size is filled
    in here (the library reads fields added after the first layout only when the stated size covers them)."""
    _fields_ = [("d_resids", C.c_void_p), ("resid_runs", C.c_int32), ("struct_size", C.c_int32), ("d_hist", C.c_void_p), ("d_screen", C.c_void_p)]

    def __init__(self, d_resids=None, resid_runs=0, struct_size=0, d_hist=None, d_screen=None):
        super().__init__(d_resids, int(resid_runs), int(struct_size) or C.sizeof(Diag), d_hist, d_screen)


class H2elParams(C.Structure):
    _fields_ = [("th", C.c_double), ("conf", C.c_double), ("max_iters", C.c_int32), ("do_lo", C.c_int32),
                ("inl_limit", C.c_int32), ("reserved", C.c_int32)]


class MiDegensacError(RuntimeError):
    pass


_lib = None


def _preload_torch_hip_runtime():
    """PyTorch-ROCm wheels bundle their own libamdhip64.so.7 / libhsa-runtime64.so.1 (same SONAMEs as /opt/rocm's).
    Whichever copy is mapped first serves the whole process; if /opt/rocm's comes first, a later `import torch` finds
    "No HIP GPUs".  So when torch is installed, map ITS runtime before libmi_degensac.so (without importing torch:
    that costs seconds); the library then runs on that runtime, exactly as when the caller imported torch first."""
    import importlib.util
    import sys
    if "torch" in sys.modules or os.environ.get("MI_DEGENSAC_NO_TORCH_RUNTIME"):
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return
    libdir = os.path.join(list(spec.submodule_search_locations)[0], "lib")
    for name in ("libhsa-runtime64.so", "libamdhip64.so"):
        path = os.path.join(libdir, name)
        if os.path.exists(path):
            try:
                C.CDLL(path, mode=C.RTLD_GLOBAL)
            except OSError:
                return


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise MiDegensacError(f"{LIB_PATH} is not built (run `python -c 'import __graft_entry__ as g; g.build()'`); "
                                  "there is no CPU fallback")
        _preload_torch_hip_runtime()
        l = C.CDLL(LIB_PATH)
        dp = C.POINTER(C.c_double); ip = C.POINTER(C.c_int32); up = C.POINTER(C.c_uint32); bp = C.POINTER(C.c_uint8)
        lp = C.POINTER(C.c_int64); pp = C.POINTER(Params)
        for name in ("mi_degensac_find_fundamental", "mi_degensac_find_homography"):
            f = getattr(l, name); f.restype = C.c_int
            f.argtypes = [dp, dp, C.c_int, C.c_int, pp, C.c_uint32, C.c_int, dp, bp, ip]
        for name in ("mi_degensac_find_fundamental_batch", "mi_degensac_find_homography_batch"):
            f = getattr(l, name); f.restype = C.c_int
            f.argtypes = [dp, dp, lp, C.c_int, C.c_int, pp, up, C.c_int, dp, bp, ip]
        for name in ("mi_degensac_find_fundamental_batch_multi", "mi_degensac_find_homography_batch_multi"):
            f = getattr(l, name); f.restype = C.c_int
            f.argtypes = [dp, dp, lp, C.c_int, C.c_int, pp, up, ip, C.c_int, dp, bp, ip]
        l.mi_degensac_set_wait_ticks.restype = C.c_longlong
        l.mi_degensac_set_wait_ticks.argtypes = [C.c_longlong]
        l.mi_degensac_ransac_h2el_batch.restype = C.c_int
        l.mi_degensac_ransac_h2el_batch.argtypes = [dp, lp, C.c_int, C.POINTER(H2elParams), up, C.c_int, dp, bp, ip]
        l.mi_degensac_ransac_h2el_batch_dev.restype = C.c_int
        l.mi_degensac_ransac_h2el_batch_dev.argtypes = [C.c_void_p, C.c_void_p, lp, C.c_int, C.POINTER(H2elParams), C.c_void_p, C.c_int,
                                                        C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        for name in ("mi_degensac_find_fundamental_batch_dev", "mi_degensac_find_homography_batch_dev"):
            f = getattr(l, name); f.restype = C.c_int
            f.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, lp, C.c_int, C.c_int, pp, C.c_void_p, C.c_int, C.c_void_p,
                          C.c_void_p, C.c_void_p, C.c_void_p]
        for name in ("mi_degensac_find_fundamental_batch_dev_ex", "mi_degensac_find_homography_batch_dev_ex"):
            f = getattr(l, name); f.restype = C.c_int
            f.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, lp, C.c_int, C.c_int, pp, C.c_void_p, C.c_int, C.c_void_p,
                          C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(Diag)]
        for name in ("mi_degensac_ctx_find_fundamental_batch", "mi_degensac_ctx_find_homography_batch"):
            f = getattr(l, name); f.restype = C.c_int
            f.argtypes = [C.c_void_p, dp, dp, lp, C.c_int, C.c_int, pp, up, dp, bp, ip]
        l.mi_degensac_ctx_create.restype = C.c_int
        l.mi_degensac_ctx_create.argtypes = [C.c_int, C.POINTER(C.c_void_p)]
        l.mi_degensac_ctx_destroy.restype = None
        l.mi_degensac_ctx_destroy.argtypes = [C.c_void_p]
        l.mi_degensac_ctx_stream.restype = C.c_void_p
        l.mi_degensac_ctx_stream.argtypes = [C.c_void_p]
        if hasattr(l, "mi_degensac_ctx_set_scheduling"):          # (absent from older builds loaded through MI_DEGENSAC_LIB for A/B runs)
            l.mi_degensac_ctx_set_scheduling.restype = C.c_int
            l.mi_degensac_ctx_set_scheduling.argtypes = [C.c_void_p, C.c_int, C.c_int]
        l.mi_degensac_release_scratch.restype = C.c_int
        l.mi_degensac_release_scratch.argtypes = [C.c_int, C.c_void_p]
        l.mi_degensac_pool_stage_parallel.restype = C.c_int
        l.mi_degensac_pool_stage_parallel.argtypes = [C.c_int]
        l.mi_degensac_sample_stream_ex.restype = C.c_int
        l.mi_degensac_sample_stream_ex.argtypes = [C.c_uint32, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, ip]
        l.mi_degensac_score_models.restype = C.c_int
        l.mi_degensac_score_models.argtypes = [dp, dp, C.c_int, C.c_int, dp, C.c_int, C.c_int, C.c_double, C.c_int, up, dp, dp]
        l.mi_degensac_sample_stream.restype = C.c_int
        l.mi_degensac_sample_stream.argtypes = [C.c_uint32, C.c_int, C.c_int, C.c_int, C.c_int, ip]
        l.mi_degensac_solve7.restype = C.c_int
        l.mi_degensac_solve7.argtypes = [dp, dp, C.c_int, C.c_int, ip, C.c_int, C.c_int, ip, ip, dp]
        for name in ("mi_degensac_find_fundamental_resids", "mi_degensac_find_homography_resids"):
            f = getattr(l, name); f.restype = C.c_int
            f.argtypes = [dp, dp, C.c_int, C.c_int, pp, C.c_uint32, C.c_int, dp, bp, ip, dp, C.c_int]
        l.mi_degensac_find_fundamental_hist.restype = C.c_int
        l.mi_degensac_find_fundamental_hist.argtypes = [dp, dp, C.c_int, C.c_int, pp, C.c_uint32, C.c_int, dp, bp, ip, ip]
        l.mi_degensac_match.restype = C.c_int
        l.mi_degensac_match.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_int, C.c_int, ip,
                                        C.POINTER(C.c_float), bp]
        l.mi_degensac_match_knn2_dev.restype = C.c_int
        l.mi_degensac_match_knn2_dev.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        l.mi_degensac_match_filter_dev.restype = C.c_int
        l.mi_degensac_match_filter_dev.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_float, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        l.mi_degensac_kpts_to_xyA.restype = C.c_int
        l.mi_degensac_kpts_to_xyA.argtypes = [C.POINTER(C.c_float), C.c_int, C.c_int, dp]
        l.mi_degensac_kpts_to_xyA_dev.restype = C.c_int
        l.mi_degensac_kpts_to_xyA_dev.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        l.mi_degensac_match_last_error.restype = C.c_char_p
        l.mi_degensac_mat3.restype = C.c_int
        l.mi_degensac_mat3.argtypes = [C.c_int, dp, C.c_int, C.c_int, dp, ip]
        l.mi_degensac_screen_counts.restype = C.c_int
        l.mi_degensac_screen_counts.argtypes = [dp, dp, C.c_int, C.c_int, dp, C.c_int, C.c_int, C.c_double, C.c_int, up, up]
        l.mi_degensac_screen_counts_h.restype = C.c_int
        l.mi_degensac_screen_counts_h.argtypes = [dp, dp, C.c_int, C.c_int, dp, C.c_int, C.c_double, C.c_int, up, bp]
        l.mi_degensac_rng_wave.restype = C.c_int
        l.mi_degensac_rng_wave.argtypes = [C.c_uint32, C.c_int, C.c_int, C.c_int, C.c_int, ip]
        if hasattr(l, "mi_degensac_set_call_timing"):
            l.mi_degensac_set_call_timing.restype = C.c_int
            l.mi_degensac_set_call_timing.argtypes = [C.c_int]
            l.mi_degensac_last_call_timing.restype = C.c_int
            l.mi_degensac_last_call_timing.argtypes = [dp]
        l.mi_degensac_last_error.restype = C.c_char_p
        l.mi_degensac_version.restype = C.c_char_p
        l.mi_degensac_kernel_name.restype = C.c_char_p
        l.mi_degensac_device_count.restype = C.c_int
        _lib = l
    return _lib


def check(rc):
    if rc != 0:
        msg = lib().mi_degensac_last_error().decode()
        if rc == -1:
            raise ValueError(msg)          # std::invalid_argument -> ValueError in the reference binding
        raise MiDegensacError(f"mi_degensac error {rc}: {msg}")


def dptr(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def make_params(px_th, conf, max_iters, error_type, sym_check, laf_coef, degen=True, flags=0, tuning=0):
    return Params(float(px_th), float(conf), int(max_iters), int(error_type), int(bool(sym_check)), int(bool(degen)),
                  float(laf_coef), int(flags), int(tuning))


def set_stream_mode(mode):
    """stream mode of the fundamental-matrix kernel (include/mi_degensac.h): -1 automatic, 0 off, odd > 0 = on with test bits; returns the previous mode"""
    f = lib().mi_degensac_set_stream_mode
    f.restype = C.c_int; f.argtypes = [C.c_int]
    return int(f(int(mode)))


def set_hjob_mode(mode):
    """homography helper workgroups (include/mi_degensac.h): 1 on, 0 off; returns the previous mode"""
    f = lib().mi_degensac_set_hjob_mode
    f.restype = C.c_int; f.argtypes = [C.c_int]
    return int(f(int(mode)))


def set_wait_ticks(ticks):
    """test hook: limit of the stream mode's data waits in 100 MHz device ticks (0: every data wait fails at once; < 0: the default
    4 s); returns the previous limit"""
    return int(lib().mi_degensac_set_wait_ticks(int(ticks)))


TIMING_NAMES = ["call_ms", "pack_ms", "enqueue_ms", "wait_ms", "unpack_ms", "dev_h2d_ms", "dev_kernel_ms", "dev_d2h_ms"]


def set_call_timing(on):
    """per-call timing of the host-pointer entry points (include/mi_degensac.h); returns the previous switch"""
    return int(lib().mi_degensac_set_call_timing(int(bool(on))))


def last_call_timing():
    """the calling thread's last timed host-pointer call: dict of TIMING_NAMES (milliseconds)"""
    out = (C.c_double * len(TIMING_NAMES))()
    check(lib().mi_degensac_last_call_timing(out))
    return {k: float(v) for k, v in zip(TIMING_NAMES, out)}


def stats_dict(st):
    d = {k: int(v) for k, v in zip(STAT_NAMES, st)}
    d["set_aside"] = (d["placement"] >> 8) & 1    # the pair was written back to its workspace once and resumed later
    d["streamed"] = (d["placement"] >> 9) & 1     # the pair took chunks from a producer workgroup (stream mode)
    d["discarded"] = (d["placement"] >> 10) & 1   # a hand-over wait of the launch timed out: results discarded (zero model / mask)
    d["rerun"] = (d["placement"] >> 11) & 1       # ... and the host-pointer entry point ran the pair again without helpers
    d["placement"] &= 255
    return d
