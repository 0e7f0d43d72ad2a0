"""TEST INFRASTRUCTURE ONLY (never imported by pydegensac_amd/): numpy restatement of the matcher stage the reference's
example runs through OpenCV (examples/simple-example.py:46-53: BFMatcher().knnMatch(descs1, descs2, k=2) and the ratio
test `m.distance < 0.9 * n.distance`).  OpenCV (cv2, absent from this image) is the third-party dependency; its
brute-force matcher returns, per query, the train rows in ascending distance, distance = NORM_L2 of the float32
difference (NORM_HAMMING for uint8), first-found on ties.  Stated here with one fixed arithmetic so that the GPU
kernel can be compared bit for bit: fp32 squared differences accumulated over the dimension in ascending order."""
import numpy as np


def dist_matrix(desc1, desc2, norm="l2"):
    a = np.asarray(desc1); b = np.asarray(desc2)
    if norm == "l2":
        a = a.astype(np.float32); b = b.astype(np.float32)
        acc = np.zeros((a.shape[0], b.shape[0]), np.float32)
        for k in range(a.shape[1]):                              # ascending k, one rounding per operation
            d = (a[:, k, None] - b[None, :, k]).astype(np.float32)
            acc = (acc + (d * d).astype(np.float32)).astype(np.float32)
        return np.sqrt(acc).astype(np.float32)
    x = np.bitwise_xor(a[:, None, :], b[None, :, :])
    return np.unpackbits(x, axis=2).sum(axis=2).astype(np.float32)


def knn2(desc1, desc2, norm="l2"):
    D = dist_matrix(desc1, desc2, norm)
    n1, n2 = D.shape
    idx = np.full((n1, 2), -1, np.int32); dist = np.full((n1, 2), np.inf, np.float32)
    order = np.argsort(D, axis=1, kind="stable")[:, :2]          # stable: the lower index wins a tie
    for k in range(min(2, n2)):
        idx[:, k] = order[:, k]; dist[:, k] = D[np.arange(n1), order[:, k]]
    return idx, dist


def match_snn(desc1, desc2, ratio=0.9, mutual=False, norm="l2"):
    idx, dist = knn2(desc1, desc2, norm)
    keep = (idx[:, 1] >= 0) & (dist[:, 0] < np.float32(ratio) * dist[:, 1])
    if mutual:
        back, _ = knn2(desc2, desc1, norm)
        keep &= back[np.clip(idx[:, 0], 0, None), 0] == np.arange(len(idx))
    q = np.flatnonzero(keep)
    return q, idx[q, 0].astype(np.int64), dist[q, 0]
