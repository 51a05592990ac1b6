"""Re-creation of the reference's only numerical self-check (deepim/operator_py/transform3d.py:365-638) on the
CPU oracle: forward vs RT_transform + matmul (|d| < 1e-4, :473-493), backward vs finite differences
(step 1e-2 / thresh 5e-3, :503-638).  Seed 1, B=8, N=3000, v_pose_src as :430-432."""
import numpy as np

from oracle import se3 as ose3, transform3d as ot3d


def _inputs():
    np.random.seed(1)
    B, N = 8, 3000
    pts = (np.random.rand(B, 3, N).astype(np.float32) - 0.5) * 0.2
    rot = np.random.rand(B, 4).astype(np.float32) - 0.5
    rot /= np.linalg.norm(rot, axis=1, keepdims=True)
    trans = (np.random.rand(B, 3).astype(np.float32) - 0.5) * 0.2
    pose_src = np.tile(np.array([[0, 1, 0, 0], [1, 0, 0, 0], [0, 0, 1, 1]], dtype=np.float32), (B, 1, 1))
    return pts, rot.astype(np.float32), trans, pose_src


def test_forward_matches_RT_transform():
    pts, rot, trans, pose_src = _inputs()
    z3, o3 = np.zeros(3, np.float32), np.ones(3, np.float32)
    for coord in ["MODEL", "CAMERA", "CAMERA_NEW", "NAIVE"]:
        out = ot3d.forward(pts, rot, trans, pose_src, z3, o3, coord)
        for b in range(pts.shape[0]):
            pose = ose3.RT_transform(pose_src[b], rot[b], trans[b], z3, o3, coord)
            ref = pose[:, :3] @ pts[b] + pose[:, 3:4]
            assert np.abs(out[b] - ref).max() < 1e-4


def test_backward_matches_finite_differences():
    pts, rot, trans, pose_src = _inputs()
    z3, o3 = np.zeros(3, np.float32), np.ones(3, np.float32)
    rng = np.random.RandomState(3)
    gout = rng.randn(*pts.shape).astype(np.float32) / pts.shape[2]
    for coord in ["MODEL", "CAMERA", "CAMERA_NEW"]:
        d_rot, d_trans = ot3d.backward(gout, pts, rot, trans, pose_src, z3, o3, coord)

        def loss(r, t):
            # differentiate through the *normalised* quaternion, as quat2mat_backward does
            rn = r / np.linalg.norm(r, axis=1, keepdims=True)
            out = np.stack([
                ose3.RT_transform(pose_src[b].astype(np.float64), rn[b], t[b], z3, o3, coord)[:, :3] @ pts[b].astype(np.float64)
                + ose3.RT_transform(pose_src[b].astype(np.float64), rn[b], t[b], z3, o3, coord)[:, 3:4] for b in range(pts.shape[0])])
            return (out * gout).sum(axis=(1, 2))

        eps = 1e-4
        for j in range(3):
            tp, tm = trans.astype(np.float64).copy(), trans.astype(np.float64).copy()
            tp[:, j] += eps
            tm[:, j] -= eps
            fd = (loss(rot.astype(np.float64), tp) - loss(rot.astype(np.float64), tm)) / (2 * eps)
            assert np.abs(fd - d_trans[:, j]).max() < 5e-3, (coord, j)
        for j in range(4):
            rp, rm = rot.astype(np.float64).copy(), rot.astype(np.float64).copy()
            rp[:, j] += eps
            rm[:, j] -= eps
            fd = (loss(rp, trans.astype(np.float64)) - loss(rm, trans.astype(np.float64))) / (2 * eps)
            assert np.abs(fd - d_rot[:, j]).max() < 5e-3, (coord, j)
