"""Pose-error functions with the reference's names (lib/utils/pose_error.py:55-147), numpy host code (evaluation is not on
the device path).  re() uses the closed form of ||logm(R_est^T R_gt)||_F / sqrt(2) = the rotation angle."""
from __future__ import print_function, division

import numpy as np
from scipy import spatial


def transform_pts_Rt(pts, R, t):
    assert pts.shape[1] == 3
    return (R.dot(pts.T) + t.reshape((3, 1))).T


def transform_pts_Rt_2d(pts, R, t, K):
    assert pts.shape[1] == 3
    pts_c_t = K.dot(R.dot(pts.T) + t.reshape((3, 1)))
    return (pts_c_t[:2] / pts_c_t[2:3]).T


def arp_2d(R_est, t_est, R_gt, t_gt, pts, K):
    """average re-projection error in 2d (pixels)"""
    return np.linalg.norm(transform_pts_Rt_2d(pts, R_est, t_est, K) - transform_pts_Rt_2d(pts, R_gt, t_gt, K), axis=1).mean()


def add(R_est, t_est, R_gt, t_gt, pts):
    """Average Distance of Model Points (Hinterstoisser et al., ACCV 2012)"""
    return np.linalg.norm(transform_pts_Rt(pts, R_est, t_est) - transform_pts_Rt(pts, R_gt, t_gt), axis=1).mean()


def adi(R_est, t_est, R_gt, t_gt, pts):
    """Average Distance of Model Points for objects with indistinguishable views (nearest neighbour)"""
    pts_est = transform_pts_Rt(pts, R_est, t_est)
    pts_gt = transform_pts_Rt(pts, R_gt, t_gt)
    nn_dists, _ = spatial.cKDTree(pts_est).query(pts_gt, k=1)
    return nn_dists.mean()


def re(R_est, R_gt):
    """rotation error in degrees"""
    assert R_est.shape == R_gt.shape == (3, 3)
    c = (np.trace(np.dot(np.transpose(R_est), R_gt)) - 1.0) / 2.0
    return np.degrees(np.arccos(np.clip(c, -1.0, 1.0)))


def te(t_est, t_gt):
    assert t_est.size == t_gt.size == 3
    return np.linalg.norm(np.asarray(t_gt).reshape(3) - np.asarray(t_est).reshape(3))


def calc_rt_dist_m(pose_src, pose_tgt):
    """lib/pair_matching/RT_transform.py:172-183: (rotation distance in degrees, translation distance)"""
    return re(pose_src[:, :3], pose_tgt[:, :3]), np.linalg.norm(pose_tgt[:, 3] - pose_src[:, 3])
