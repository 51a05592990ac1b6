"""ADD / ADI pose errors, restating /root/reference/lib/utils/pose_error.py:72-108 (oracle; test-only).
Pinned by tests/golden/pose_error_golden.npz."""
import numpy as np
from scipy import spatial


def transform_pts_Rt(pts, R, t):
    """pose_error.py:12-25."""
    assert pts.shape[1] == 3
    return (R.dot(pts.T) + t.reshape((3, 1))).T


def add(R_est, t_est, R_gt, t_gt, pts):
    e = np.linalg.norm(transform_pts_Rt(pts, R_est, t_est) - transform_pts_Rt(pts, R_gt, t_gt), axis=1).mean()
    return e


def adi(R_est, t_est, R_gt, t_gt, pts):
    pts_est = transform_pts_Rt(pts, R_est, t_est)
    pts_gt = transform_pts_Rt(pts, R_gt, t_gt)
    nn_dists, _ = spatial.cKDTree(pts_est).query(pts_gt, k=1)
    return nn_dists.mean()
