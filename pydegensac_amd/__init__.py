"""pydegensac_amd — MI355X-native LO-RANSAC / DEGENSAC with the pydegensac Python API.

Drop-in surface (reference: src/pydegensac/__init__.py:1-4, utils.py:24-146):
    findHomography, findFundamentalMatrix, convert_cv2_kpts_to_xyA,
    findHomography_, findFundamentalMatrix_   (the private pybind entry points, bindings.cpp:484-503)
Extras: `seed=` / `device=` keywords, batch entry points and per-call statistics
(`last_stats()`), the reference's older F drivers (`ransacF_legacy`: exp_ransacF / exp_ransacFcustom) and its
2-ellipse RANSAC (`ransacH2el`: ranH2el.c), see pydegensac_amd.api.
"""
from .api import (findHomography, findFundamentalMatrix, convert_cv2_kpts_to_xyA,
                  findHomography_, findFundamentalMatrix_,
                  findFundamentalMatrixBatch, findHomographyBatch, last_stats, ransacF_legacy, ransacF_legacy_batch, ransacH2el, ransacH2el_batch,
                  error_type_dict_homography, error_type_dict_fundamental)

__all__ = ["findHomography", "findFundamentalMatrix", "convert_cv2_kpts_to_xyA",
           "findHomography_", "findFundamentalMatrix_", "findFundamentalMatrixBatch",
           "findHomographyBatch", "last_stats", "ransacF_legacy", "ransacF_legacy_batch", "ransacH2el", "ransacH2el_batch"]
