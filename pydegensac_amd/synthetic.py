"""Synthetic correspondence generators for the BASELINE.json configs.

numpy only.  The same generators feed the tests, bench.py and the golden-fixture script, so a
(config, data seed) pair names one exact input everywhere.
"""
import numpy as np

IMG_W, IMG_H, FOCAL = 1000.0, 750.0, 800.0

# v_dogman H_1_6 ground truth of the reference's bundled example pair
# (values of /root/reference/examples/img/v_dogman/H_1_6, the scene fixture named by BASELINE C1/C3)
H_1_6 = np.array([[-0.036838, -0.012358, 166.5],
                  [-0.28304, 0.70203, 61.488],
                  [-0.0011124, -1.7983e-05, 0.99335]], dtype=np.float64)


def _project(K, R, t, X):
    x = (K @ (R @ X.T + t[:, None])).T
    return x[:, :2] / x[:, 2:3]


def two_view_fundamental(n=2000, inlier_ratio=0.4, sigma=0.1, seed=0, plane_fraction=0.0, laf=False, laf_bad=0.25, laf_sigma=0.05):
    """C2 / C2b / C5 generator: two pinhole views of a random 3-D cloud + uniform outliers.

    Returns (pts1 [n,2], pts2 [n,2], is_inlier [n] bool, F_gt [3,3]) with x2^T F_gt x1 = 0.
    plane_fraction > 0 puts that share of the inlier 3-D points on the plane z = 6 + 0.1 x (C2b).
    laf=True returns [n,6] rows (x, y, a11, a12, a21, a22) as findFundamentalMatrix takes them with
    laf_consistensy_coef > 0 (utils.py:111-146; the binding turns every row into the two extra points
    (x + a11, y + a21) and (x + a12, y + a22), bindings.cpp:337-409): the frame of an inlier is spanned by
    the projections of two 3-D neighbours of its point, so both extra points obey the same epipolar geometry
    (noise laf_sigma px), except for a share laf_bad of the inliers whose image-2 frame is random — their
    extra points fail the LAF check, so S.Ilafs < S.I and candidates do get rejected on it
    (exp_ranF.c:1394-1411); outliers carry random frames in both images.
    """
    rng = np.random.default_rng(seed)
    n_in = int(round(n * inlier_ratio))
    X = np.stack([rng.uniform(-2, 2, n_in), rng.uniform(-1.5, 1.5, n_in), rng.uniform(4, 8, n_in)], 1)
    if plane_fraction > 0:
        k = int(round(n_in * plane_fraction))
        X[:k, 2] = 6.0 + 0.1 * X[:k, 0]
    K = np.array([[FOCAL, 0, IMG_W / 2], [0, FOCAL, IMG_H / 2], [0, 0, 1.0]])
    a = 0.2
    R = np.array([[np.cos(a), 0, np.sin(a)], [0, 1, 0], [-np.sin(a), 0, np.cos(a)]])
    t = np.array([1.0, 0.1, 0.2])
    p1 = _project(K, np.eye(3), np.zeros(3), X) + rng.normal(0, sigma, (n_in, 2))
    p2 = _project(K, R, t, X) + rng.normal(0, sigma, (n_in, 2))
    n_out = n - n_in
    o1 = np.stack([rng.uniform(0, IMG_W, n_out), rng.uniform(0, IMG_H, n_out)], 1)
    o2 = np.stack([rng.uniform(0, IMG_W, n_out), rng.uniform(0, IMG_H, n_out)], 1)
    pts1 = np.concatenate([p1, o1]); pts2 = np.concatenate([p2, o2])
    lab = np.concatenate([np.ones(n_in, bool), np.zeros(n_out, bool)])
    perm = rng.permutation(n)
    if laf:
        # separate generator: the point coordinates and the permutation above are those of laf=False
        lrng = np.random.default_rng([seed, 0x1AF])
        A1 = np.empty((n, 4)); A2 = np.empty((n, 4))
        A1[n_in:] = lrng.normal(0, 8.0, (n_out, 4)); A2[n_in:] = lrng.normal(0, 8.0, (n_out, 4))
        for col, (k0, k1) in enumerate([(0, 2), (1, 3)]):         # column 0 of the frame = (a11, a21), column 1 = (a12, a22)
            d = lrng.normal(0, 1.0, (n_in, 3)); d[:, 2] *= 0.3
            d *= (lrng.uniform(0.04, 0.12, n_in) / np.linalg.norm(d, axis=1))[:, None]
            q1 = _project(K, np.eye(3), np.zeros(3), X + d) + lrng.normal(0, laf_sigma, (n_in, 2))
            q2 = _project(K, R, t, X + d) + lrng.normal(0, laf_sigma, (n_in, 2))
            A1[:n_in, k0] = q1[:, 0] - p1[:, 0]; A1[:n_in, k1] = q1[:, 1] - p1[:, 1]
            A2[:n_in, k0] = q2[:, 0] - p2[:, 0]; A2[:n_in, k1] = q2[:, 1] - p2[:, 1]
        bad = lrng.random(n_in) < laf_bad
        A2[:n_in][bad] = lrng.normal(0, 8.0, (int(bad.sum()), 4))
        pts1 = np.concatenate([pts1, A1], 1); pts2 = np.concatenate([pts2, A2], 1)
    tx = np.array([[0, -t[2], t[1]], [t[2], 0, -t[0]], [-t[1], t[0], 0]])
    Kinv = np.linalg.inv(K)
    F = Kinv.T @ tx @ R @ Kinv
    return (np.ascontiguousarray(pts1[perm]), np.ascontiguousarray(pts2[perm]), lab[perm], F)


def homography_pairs(n=5000, inlier_ratio=0.4, sigma=0.5, seed=0, laf=False, H=None, w=788.0, h=522.0):
    """C1 / C3 generator: points mapped through a ground-truth homography + uniform outliers.

    With laf=True the arrays are [n,6] = (x, y, a11, a12, a21, a22) with an isotropic 5 px frame
    in image 1 and the frame mapped by the local affine approximation of H in image 2.
    """
    rng = np.random.default_rng(seed)
    H = H_1_6 if H is None else H
    n_in = int(round(n * inlier_ratio))
    p1 = np.stack([rng.uniform(0, w, n_in), rng.uniform(0, h, n_in)], 1)
    ph = np.concatenate([p1, np.ones((n_in, 1))], 1) @ H.T
    p2 = ph[:, :2] / ph[:, 2:3] + rng.normal(0, sigma, (n_in, 2))
    p1 = p1 + rng.normal(0, sigma, (n_in, 2))
    n_out = n - n_in
    o1 = np.stack([rng.uniform(0, w, n_out), rng.uniform(0, h, n_out)], 1)
    o2 = np.stack([rng.uniform(0, w, n_out), rng.uniform(0, h, n_out)], 1)
    pts1 = np.concatenate([p1, o1]); pts2 = np.concatenate([p2, o2])
    lab = np.concatenate([np.ones(n_in, bool), np.zeros(n_out, bool)])
    if laf:
        A1 = np.tile(np.array([5.0, 0.0, 0.0, 5.0]), (n, 1))
        # local affine of H at each image-1 point (Jacobian of the projective map)
        q = np.concatenate([pts1, np.ones((n, 1))], 1) @ H.T
        wz = q[:, 2]
        J = np.empty((n, 2, 2))
        for r in range(2):
            for c in range(2):
                J[:, r, c] = (H[r, c] * wz - q[:, r] * H[2, c]) / (wz * wz)
        A2 = np.einsum('nij,njk->nik', J, A1.reshape(n, 2, 2)).reshape(n, 4)
        A2[~lab] = rng.normal(0, 5.0, (n_out, 4))
        pts1 = np.concatenate([pts1, A1], 1); pts2 = np.concatenate([pts2, A2], 1)
    perm = rng.permutation(n)
    return np.ascontiguousarray(pts1[perm]), np.ascontiguousarray(pts2[perm]), lab[perm], H


def ellipse_pairs(n=1000, inlier_ratio=0.3, sigma=1.0, seed=0, laf_noise=0.05, H=None, w=788.0, h=522.0):
    """Generator for ransacH2el (ranH2el.c): u10 [n, 10] = x1 y1 a1 b1 c1 | x2 y2 a2 b2 c2, one lower-triangular local
    affine frame [a 0; b c] per image.  The frame of image 2 is random, the one of image 1 is its image under the local
    affine approximation of the ground-truth homography (which maps image 2 to image 1, like the reference's internal H),
    perturbed by laf_noise and reduced to lower-triangular form by a rotation (an ellipse does not fix one).  Outliers
    keep their frames and get a uniformly random position in image 1.  Returns (u10, inlier labels)."""
    rng = np.random.default_rng(seed)
    H = H_1_6 if H is None else H
    x2 = np.stack([rng.uniform(0, w, n), rng.uniform(0, h, n)], 1)
    a = rng.uniform(5, 30, n); b = rng.uniform(-10, 10, n); c = rng.uniform(5, 30, n)
    q = np.concatenate([x2, np.ones((n, 1))], 1) @ H.T
    wz = q[:, 2]
    x1 = q[:, :2] / wz[:, None]
    J = np.empty((n, 2, 2))
    for r in range(2):
        for k in range(2):
            J[:, r, k] = (H[r, k] * wz - q[:, r] * H[2, k]) / (wz * wz)
    A2 = np.zeros((n, 2, 2)); A2[:, 0, 0] = a; A2[:, 1, 0] = b; A2[:, 1, 1] = c
    M = np.einsum('nij,njk->nik', J, A2) * (1.0 + laf_noise * rng.normal(size=(n, 2, 2)))
    r0 = np.hypot(M[:, 0, 0], M[:, 0, 1])
    Q = np.empty((n, 2, 2)); Q[:, 0, 0] = M[:, 0, 0] / r0; Q[:, 0, 1] = -M[:, 0, 1] / r0; Q[:, 1, 0] = M[:, 0, 1] / r0; Q[:, 1, 1] = M[:, 0, 0] / r0
    L = np.einsum('nij,njk->nik', M, Q)
    n_in = int(round(n * inlier_ratio))
    lab = np.zeros(n, bool); lab[:n_in] = True
    x1[:n_in] += rng.normal(0, sigma, (n_in, 2))
    x1[n_in:] = np.stack([rng.uniform(0, w, n - n_in), rng.uniform(0, h, n - n_in)], 1)
    u = np.stack([x1[:, 0], x1[:, 1], L[:, 0, 0], L[:, 1, 0], L[:, 1, 1], x2[:, 0], x2[:, 1], a, b, c], 1)
    perm = rng.permutation(n)
    return np.ascontiguousarray(u[perm]), lab[perm]
