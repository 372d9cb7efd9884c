"""CPU oracle for the point-to-point ICP refinement -- TEST INFRASTRUCTURE ONLY (see oracle/scream_ref.py).

The reference calls ``open3d.registration_icp(src, tgt, max_correspondence_distance, init)``
(evaluate_3d_match.py:109-113; KITTI: evaluate_kitti.py:64-70 with max_iteration=1000).  open3d is not in this
image and not under /root/reference (requirements.txt:2-3 pins open3d==0.17.0 / open3d_python==0.7.0.0), so
parity with open3d itself is UNPINNED.  This restates open3d's published RegistrationICP loop
(cpp/open3d/pipelines/registration/Registration.cpp): nearest target within the radius per transformed source
point; fitness = #corr / N, inlier_rmse = sqrt(mean squared distance); rigid update by Kabsch without
scaling (TransformationEstimationPointToPoint); defaults max_iteration = 30, relative_fitness =
relative_rmse = 1e-6.  float64 throughout; scipy's cKDTree stands in for the KD-tree search.
"""
import numpy as np
from scipy.spatial import cKDTree


def _kabsch(A, B):
    ca, cb = A.mean(0), B.mean(0)
    H = (A - ca).T @ (B - cb)
    U, S, Vt = np.linalg.svd(H)
    d = np.sign(np.linalg.det(Vt.T @ U.T))
    R = Vt.T @ np.diag([1.0, 1.0, d]) @ U.T
    T = np.eye(4)
    T[:3, :3] = R
    T[:3, 3] = cb - R @ ca
    return T


def icp_p2p(src, tgt, T_init, max_corr_dist, max_iter=30, rel_fitness=1e-6, rel_rmse=1e-6):
    """src [N,3], tgt [M,3] metric-frame float64.  Returns (T 4x4, fitness, inlier_rmse, n_updates)."""
    tree = cKDTree(tgt)
    T = np.array(T_init, dtype=np.float64)

    def evaluate(T):
        q = src @ T[:3, :3].T + T[:3, 3]
        d, j = tree.query(q, k=1, distance_upper_bound=max_corr_dist)
        ok = np.isfinite(d)
        n = int(ok.sum())
        return q, j, ok, (n / len(src) if len(src) else 0.0), (float(np.sqrt((d[ok] ** 2).mean())) if n else 0.0)

    q, j, ok, fit, rmse = evaluate(T)
    it = 0
    while it < max_iter:
        if ok.sum() == 0:
            break
        T = _kabsch(q[ok], tgt[j[ok]]) @ T
        it += 1
        q, j, ok, fit2, rmse2 = evaluate(T)
        conv = abs(fit - fit2) < rel_fitness and abs(rmse - rmse2) < rel_rmse
        fit, rmse = fit2, rmse2
        if conv:
            break
    return T, fit, rmse, it
