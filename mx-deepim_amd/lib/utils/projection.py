"""Rigid-transform helpers and pinhole back-projection with the reference's names (lib/utils/projection.py:12-67).

Poses are 3x4 [R|t] arrays.  Like the reference, se3_mul / se3_inverse return FLOAT32 (the labels built from them inherit that
rounding: SURVEY.md A7); backproject_camera returns a (3, H*W) float64 array of camera-frame points, row-major over pixels."""
import numpy as np


def _split(RT):
    RT = np.asarray(RT)
    return RT[:3, :3], RT[:3, 3]


def se3_inverse(RT):
    R, t = _split(RT)
    out = np.empty((3, 4), dtype=np.float32)
    out[:, :3] = R.T
    out[:, 3] = -(R.T @ t)
    return out


def se3_mul(RT1, RT2):
    (R1, t1), (R2, t2) = _split(RT1), _split(RT2)
    out = np.empty((3, 4), dtype=np.float32)
    out[:, :3] = R1 @ R2
    out[:, 3] = R1 @ t2 + t1
    return out


def backproject_camera(depth, intrinsic_matrix, FLIP_X=False):
    """X[:, v*W + u] = depth[v, u] * K^-1 [u, v, 1]^T"""
    depth = np.asarray(depth)
    H, W = depth.shape[:2]
    Kinv = np.linalg.inv(np.asarray(intrinsic_matrix, dtype=np.float64).reshape(3, 3))
    u, v = np.meshgrid(np.arange(W), np.arange(H))
    rays = Kinv @ np.stack([u.ravel(), v.ravel(), np.ones(H * W, dtype=np.float32)]).astype(np.float64)
    return rays * depth.reshape(1, H * W)
