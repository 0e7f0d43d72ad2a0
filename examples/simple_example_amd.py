#!/usr/bin/env python
"""The reference's examples/simple-example.py on this package: same call pattern (`import pydegensac`, tentative
correspondences from a 2-NN ratio test, then findHomography / findFundamentalMatrix), with the two OpenCV stages that
are not available offline replaced — detection by synthetic keypoints + descriptors, matching by the GPU matcher."""
import os
import sys
from time import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))    # run from a checkout

import pydegensac                                    # the alias package: everything runs on the MI355X
from pydegensac_amd import matcher, synthetic


def verify_pydegensac(kps1, kps2, tentatives, th=4.0, n_iter=2000):
    src_pts = np.float32([kps1[q] for q, t in tentatives]).reshape(-1, 2)
    dst_pts = np.float32([kps2[t] for q, t in tentatives]).reshape(-1, 2)
    H, mask = pydegensac.findHomography(src_pts, dst_pts, th, 0.99, n_iter)
    print('pydegensac found {} inliers'.format(int(np.asarray(mask, np.float32).sum())))
    return H, mask


def verify_pydegensac_fundam(kps1, kps2, tentatives, th=1.0, n_iter=10000):
    src_pts = np.float32([kps1[q] for q, t in tentatives]).reshape(-1, 2)
    dst_pts = np.float32([kps2[t] for q, t in tentatives]).reshape(-1, 2)
    F, mask = pydegensac.findFundamentalMatrix(src_pts, dst_pts, th, 0.999, n_iter, enable_degeneracy_check=True)
    print('pydegensac found {} inliers'.format(int(np.asarray(mask, np.float32).sum())))
    return F, mask


if __name__ == '__main__':
    rng = np.random.default_rng(0)
    # "detector": two views of a plane, keypoints with descriptors (inliers share a noisy descriptor)
    kps1, kps2, lab, _ = synthetic.homography_pairs(n=3000, inlier_ratio=0.4, sigma=0.5, seed=1)
    descs1 = rng.normal(size=(3000, 64)).astype(np.float32)
    descs2 = descs1 + 0.15 * rng.normal(size=descs1.shape).astype(np.float32)
    descs2[~lab] = rng.normal(size=((~lab).sum(), 64)).astype(np.float32)
    # bf = cv2.BFMatcher(); matches = bf.knnMatch(descs1, descs2, k=2); SNN ratio test m.distance < 0.9 * n.distance
    q, t, _ = matcher.match_snn(descs1, descs2, ratio=0.9)
    tentatives = list(zip(q, t))
    print(len(tentatives), 'tentative correspondences')
    t0 = time(); H, mask = verify_pydegensac(kps1, kps2, tentatives, 4.0, 2000)
    print("pydegensac runtime {0:.5f}".format(time() - t0), ' sec'); print("H = ", H)
    t0 = time(); F, mask = verify_pydegensac_fundam(kps1, kps2, tentatives, 0.5, 50000)
    print("pydegensac {0:.5f}".format(time() - t0), ' sec'); print("F = ", F)
