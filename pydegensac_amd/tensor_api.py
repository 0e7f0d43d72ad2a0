"""Device-resident batch API: correspondences and results stay in HBM.

The natural caller of a robust estimator is a matching pipeline that already holds its tentative
correspondences on the GPU.  These functions take `torch` tensors on a ROCm device (anything exposing
`data_ptr()`, float64, C-contiguous), run the same persistent kernels through the `*_batch_dev` entry points
of include/mi_degensac.h on the tensor's current stream, and return tensors — no host staging, no
synchronisation.  Marshalling rules follow bindings.cpp:126-198 (rows are [x, y] or [x, y, a11, a12, a21,
a22]); seeds follow pydegensac_amd.parallel.pair_seeds unless given (any uint32 value is allowed).
"""
import ctypes as C

import numpy as np

from . import _lib
from .api import error_type_dict_fundamental, error_type_dict_homography


def _prep(pts1, pts2, counts):
    import torch
    if not (isinstance(pts1, torch.Tensor) and isinstance(pts2, torch.Tensor)):
        raise ValueError("pts1/pts2 must be torch tensors on the GPU")
    if pts1.device.type != "cuda" or pts2.device != pts1.device:
        raise ValueError("pts1/pts2 must live on the same ROCm device")
    if pts1.dtype != torch.float64 or pts2.dtype != torch.float64:
        raise ValueError("correspondences must be float64 (the path is fp64 end to end)")
    if pts1.dim() != 2 or pts1.shape != pts2.shape or pts1.shape[1] not in (2, 6):
        raise ValueError("expected two [total, 2] or [total, 6] tensors of equal shape")
    counts = np.asarray(counts, dtype=np.int64).ravel()
    offs = np.zeros(len(counts) + 1, dtype=np.int64); np.cumsum(counts, out=offs[1:])
    if offs[-1] != pts1.shape[0]:
        raise ValueError("counts do not add up to the number of rows")
    return pts1.contiguous(), pts2.contiguous(), offs


def _run(which, pts1, pts2, counts, prm, seeds, min_n):
    import torch
    from . import parallel
    pts1, pts2, offs = _prep(pts1, pts2, counts)
    P = len(offs) - 1
    if P == 0 or (np.diff(offs) < min_n).any():
        raise ValueError(f"every pair needs at least {min_n} correspondences")
    dev = pts1.device
    if seeds is None:
        seeds = parallel.pair_seeds(0, P)
    # uint32 seeds travel as their int32 bit pattern (torch has no uint32 arithmetic; the kernel reads them as unsigned)
    d_seeds = torch.from_numpy((np.asarray(seeds, dtype=np.int64) & 0xFFFFFFFF).astype(np.uint32).view(np.int32)).to(dev)
    d_off = torch.from_numpy(offs).to(dev)
    model = torch.zeros((P, 9), dtype=torch.float64, device=dev)
    mask = torch.zeros(int(offs[-1]), dtype=torch.uint8, device=dev)
    stats = torch.zeros((P, 16), dtype=torch.int32, device=dev)
    stream = torch.cuda.current_stream(dev)
    fn = _lib.lib().mi_degensac_find_fundamental_batch_dev if which == "F" else _lib.lib().mi_degensac_find_homography_batch_dev
    rc = fn(pts1.data_ptr(), pts2.data_ptr(), d_off.data_ptr(), offs.ctypes.data_as(C.POINTER(C.c_int64)), P, int(pts1.shape[1]),
            C.byref(prm), d_seeds.data_ptr(), dev.index or 0, C.c_void_p(stream.cuda_stream),
            model.data_ptr(), mask.data_ptr(), stats.data_ptr())
    _lib.check(rc)
    # the kernel reads d_off / d_seeds asynchronously: keep them alive until the stream reaches this point
    for t in (d_off, d_seeds, pts1, pts2):
        t.record_stream(stream)
    return model.view(P, 3, 3), mask.to(torch.bool), stats, offs


def find_fundamental_batch_tensors(pts1, pts2, counts, px_th=0.5, conf=0.9999, max_iters=100000,
                                   laf_consistensy_coef=-1.0, error_type="sampson", symmetric_error_check=True,
                                   enable_degeneracy_check=True, seeds=None):
    """P independent pairs, rows of pair p = pts[offs[p]:offs[p+1]] with offs = cumsum(counts).
    Returns (F [P,3,3] float64, mask [total] bool, stats [P,16] int32, offsets) — all but offsets on the device."""
    et = error_type_dict_fundamental[error_type.lower()]
    prm = _lib.make_params(px_th, conf, max_iters, et, symmetric_error_check, max(0.0, laf_consistensy_coef), enable_degeneracy_check)
    return _run("F", pts1, pts2, counts, prm, seeds, 8)


def find_homography_batch_tensors(pts1, pts2, counts, px_th=1.0, conf=0.999, max_iters=50000, laf_consistensy_coef=-1.0,
                                  error_type="sampson", symmetric_error_check=True, seeds=None):
    """As above for homographies.  Returns the reference's user-facing H = inv(H_c^T) (utils.py:108), zeros when no
    model was found (the inversion runs on the device with torch.linalg)."""
    import torch
    et = error_type_dict_homography[error_type.lower()]
    prm = _lib.make_params(px_th, conf, max_iters, et, symmetric_error_check, max(0.0, laf_consistensy_coef), True)
    Hc, mask, stats, offs = _run("H", pts1, pts2, counts, prm, seeds, 4)
    found = Hc.abs().sum(dim=(1, 2)) != 0
    out = torch.zeros_like(Hc)
    if bool(found.any()):
        out[found] = torch.linalg.inv(Hc[found].transpose(1, 2))
    return out, mask, stats, offs
