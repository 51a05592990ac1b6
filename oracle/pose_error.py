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


def transform_pts_Rt_2d(pts, R, t, K):
    """pose_error.py:28-47: rigid transform, then the pinhole projection."""
    assert pts.shape[1] == 3
    pc = K.dot(R.dot(pts.T) + t.reshape((3, 1)))
    out = np.zeros((pts.shape[0], 2))
    out[:, 0] = pc[0, :] / pc[2, :]
    out[:, 1] = pc[1, :] / pc[2, :]
    return out


def arp_2d(R_est, t_est, R_gt, t_gt, pts, K):
    """pose_error.py:50-64: mean 2-D distance of the projected model points."""
    return np.linalg.norm(transform_pts_Rt_2d(pts, R_est, t_est, K) - transform_pts_Rt_2d(pts, R_gt, t_gt, K), axis=1).mean()


def re(R_est, R_gt):
    """pose_error.py:128-134: geodesic angle through the matrix logarithm, degrees."""
    from scipy.linalg import logm

    assert R_est.shape == R_gt.shape == (3, 3)
    return np.linalg.norm(logm(np.dot(np.transpose(R_est), R_gt)), "fro") / np.sqrt(2) / np.pi * 180
