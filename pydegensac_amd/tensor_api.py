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
    Returns (F [P,3,3] float64, mask [total] bool, stats [P,16] int32, offsets) — all but offsets on the device.
    stats columns = include/mi_degensac.h MI_ST_* (column 15: placement in bits 0-7, bit 8 = the pair was set aside once)."""
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


# ---- the stage in front of the estimators on the device: matcher and keypoint conversion -------------
def knn_match_tensors(desc1, desc2):
    """cv2 `BFMatcher().knnMatch(descs1, descs2, k=2)` (examples/simple-example.py:46-47) on device tensors: float32
    descriptors [n, dim] -> L2, uint8 [n, dim] (dim % 4 == 0) -> Hamming.  Returns (idx [n1, 2] int32, dist [n1, 2] float32)
    on the device, asynchronous on the current stream."""
    import torch
    if not (isinstance(desc1, torch.Tensor) and isinstance(desc2, torch.Tensor)) or desc1.device.type != "cuda" or desc2.device != desc1.device:
        raise ValueError("descriptors must be torch tensors on the same ROCm device")
    if desc1.dim() != 2 or desc2.dim() != 2 or desc1.shape[1] != desc2.shape[1] or desc1.dtype != desc2.dtype:
        raise ValueError("descriptors should be [n1, dim] and [n2, dim] tensors of one dtype")
    if desc1.dtype == torch.float32:
        norm = 0
    elif desc1.dtype == torch.uint8 and desc1.shape[1] % 4 == 0:
        norm = 1
    else:
        raise ValueError("float32 descriptors (L2) or uint8 descriptors with dim % 4 == 0 (Hamming)")
    a = desc1.contiguous(); b = desc2.contiguous(); dev = a.device
    # the kernel reads descriptor rows as 32-bit words: a contiguous view into a packed buffer may start at any byte
    if a.data_ptr() % 4: a = a.clone()
    if b.data_ptr() % 4: b = b.clone()
    n1, n2, dim = a.shape[0], b.shape[0], a.shape[1]
    idx = torch.full((n1, 2), -1, dtype=torch.int32, device=dev)
    dist = torch.full((n1, 2), float("inf"), dtype=torch.float32, device=dev)
    stream = torch.cuda.current_stream(dev)
    rc = _lib.lib().mi_degensac_match_knn2_dev(norm, a.data_ptr(), n1, b.data_ptr(), n2, dim, dev.index or 0, C.c_void_p(stream.cuda_stream),
                                               idx.data_ptr(), dist.data_ptr())
    if rc != 0:
        raise _lib.MiDegensacError(f"mi_degensac error {rc}: {_lib.lib().mi_degensac_match_last_error().decode()}")
    for t in (a, b):
        t.record_stream(stream)
    return idx, dist


def match_snn_tensors(desc1, desc2, ratio=0.9, mutual=False):
    """The ratio test of the example (`m.distance < ratio * n.distance`, simple-example.py:49-53), optionally restricted to
    mutual nearest neighbours, on the device: (query indices, train indices, distances) of the tentative correspondences.
    The only host synchronisation is the final boolean selection (the number of survivors sizes the outputs)."""
    import torch
    idx, dist = knn_match_tensors(desc1, desc2)
    dev = idx.device; n1 = idx.shape[0]
    keep = torch.zeros(n1, dtype=torch.uint8, device=dev)
    back = knn_match_tensors(desc2, desc1)[0] if mutual and desc2.shape[0] > 0 else None
    stream = torch.cuda.current_stream(dev)
    rc = _lib.lib().mi_degensac_match_filter_dev(idx.data_ptr(), dist.data_ptr(), n1, float(ratio), back.data_ptr() if back is not None else None,
                                                 dev.index or 0, C.c_void_p(stream.cuda_stream), keep.data_ptr())
    if rc != 0:
        raise _lib.MiDegensacError(f"mi_degensac error {rc}: {_lib.lib().mi_degensac_match_last_error().decode()}")
    if back is not None:
        back.record_stream(stream)
    sel = torch.nonzero(keep, as_tuple=False).flatten()
    return sel, idx[sel, 0].to(torch.int64), dist[sel, 0]


def kpts_to_xyA_tensors(kpts):
    """utils.py:24-41 `convert_cv2_kpts_to_xyA` on the device: [n, 4] float32 (x, y, size, angle in degrees) -> [n, 6] float64"""
    import torch
    if not isinstance(kpts, torch.Tensor) or kpts.device.type != "cuda" or kpts.dtype != torch.float32 or kpts.dim() != 2 or kpts.shape[1] != 4:
        raise ValueError("keypoints should be a float32 tensor [n, 4] = (x, y, size, angle) on a ROCm device")
    k = kpts.contiguous(); dev = k.device
    out = torch.zeros((k.shape[0], 6), dtype=torch.float64, device=dev)
    stream = torch.cuda.current_stream(dev)
    rc = _lib.lib().mi_degensac_kpts_to_xyA_dev(k.data_ptr(), k.shape[0], dev.index or 0, C.c_void_p(stream.cuda_stream), out.data_ptr())
    if rc != 0:
        raise _lib.MiDegensacError(f"mi_degensac error {rc}: {_lib.lib().mi_degensac_match_last_error().decode()}")
    k.record_stream(stream)
    return out
