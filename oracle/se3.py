"""SE(3) pose algebra of DeepIM, float64 numpy restatement (oracle; test-only).

Follows /root/reference/lib/pair_matching/RT_transform.py and
lib/utils/projection.py; each function cites the lines it restates.
Pinned by tests/golden/se3_golden.npz (generated from the reference itself).
"""
import numpy as np

_FLOAT_EPS = np.finfo(np.float64).eps  # RT_transform.py:247 (np.finfo(np.float).eps)


def quat2mat(q):
    """(w,x,y,z) -> 3x3, un-normalised quats allowed. RT_transform.py:393-443."""
    w, x, y, z = [float(v) for v in q]
    Nq = w * w + x * x + y * y + z * z
    if Nq < _FLOAT_EPS:
        return np.eye(3)
    s = 2.0 / Nq
    X, Y, Z = x * s, y * s, z * s
    wX, wY, wZ = w * X, w * Y, w * Z
    xX, xY, xZ = x * X, x * Y, x * Z
    yY, yZ, zZ = y * Y, y * Z, z * Z
    return np.array(
        [
            [1.0 - (yY + zZ), xY - wZ, xZ + wY],
            [xY + wZ, 1.0 - (xX + zZ), yZ - wX],
            [xZ - wY, yZ + wX, 1.0 - (xX + yY)],
        ]
    )


def mat2quat(M):
    """3x3 -> (w,x,y,z) with w >= 0 (Bar-Itzhack eigen method). RT_transform.py:446-523."""
    Qxx, Qyx, Qzx, Qxy, Qyy, Qzy, Qxz, Qyz, Qzz = np.asarray(M, dtype=np.float64).flat
    K = (
        np.array(
            [
                [Qxx - Qyy - Qzz, 0, 0, 0],
                [Qyx + Qxy, Qyy - Qxx - Qzz, 0, 0],
                [Qzx + Qxz, Qzy + Qyz, Qzz - Qxx - Qyy, 0],
                [Qyz - Qzy, Qzx - Qxz, Qxy - Qyx, Qxx + Qyy + Qzz],
            ]
        )
        / 3.0
    )
    vals, vecs = np.linalg.eigh(K)
    q = vecs[[3, 0, 1, 2], np.argmax(vals)]
    if q[0] < 0:
        q = -q
    return q


def se3_inverse(RT):
    """projection.py:12-23 (returns float32, as the reference does)."""
    R = RT[0:3, 0:3]
    T = RT[0:3, 3].reshape((3, 1))
    out = np.zeros((3, 4), dtype=np.float32)
    out[0:3, 0:3] = R.transpose()
    out[0:3, 3] = -1 * np.dot(R.transpose(), T).reshape(3)
    return out


def se3_mul(RT1, RT2):
    """projection.py:26-43 (returns float32)."""
    R1, T1 = RT1[0:3, 0:3], RT1[0:3, 3].reshape((3, 1))
    R2, T2 = RT2[0:3, 0:3], RT2[0:3, 3].reshape((3, 1))
    out = np.zeros((3, 4), dtype=np.float32)
    out[0:3, 0:3] = np.dot(R1, R2)
    out[0:3, 3] = (np.dot(R1, T2) + T1).reshape(3)
    return out


def R_transform(R_src, R_delta, rot_coord="MODEL"):
    """RT_transform.py:51-69."""
    rc = rot_coord.lower()
    if rc == "model":
        return np.dot(R_src, R_delta)
    if rc in ("camera", "naive", "camera_new"):
        return np.dot(R_delta, R_src)
    raise Exception("Unknown rot_coord in R_transform: {}".format(rot_coord))


def R_inv_transform(R_src, R_tgt, rot_coord):
    """RT_transform.py:72-79."""
    rc = rot_coord.lower()
    if rc == "model":
        return np.dot(R_src.transpose(), R_tgt)
    if rc in ("camera", "camera_new"):
        return np.dot(R_tgt, R_src.transpose())
    raise Exception("Unknown rot_coord in R_inv_transform: {}".format(rot_coord))


def T_transform(T_src, T_delta, T_means, T_stds, rot_coord):
    """DeepIM untangled translation update. RT_transform.py:82-103."""
    assert T_src[2] != 0, "T_src: {}".format(T_src)
    d = np.asarray(T_delta, dtype=np.float64) * T_stds + T_means
    out = np.zeros((3,))
    z2 = T_src[2] / np.exp(d[2])
    out[2] = z2
    rc = rot_coord.lower()
    if rc in ("camera", "model"):
        out[0] = z2 * (d[0] + T_src[0] / T_src[2])
        out[1] = z2 * (d[1] + T_src[1] / T_src[2])
    elif rc == "camera_new":
        out[0] = T_src[2] * d[0] + T_src[0]
        out[1] = T_src[2] * d[1] + T_src[1]
    else:
        raise Exception("Unknown: {}".format(rot_coord))
    return out


def T_inv_transform(T_src, T_tgt, T_means, T_stds, rot_coord):
    """RT_transform.py:113-132."""
    d = np.zeros((3,))
    rc = rot_coord.lower()
    if rc == "camera_new":
        d[0] = (T_tgt[0] - T_src[0]) / T_src[2]
        d[1] = (T_tgt[1] - T_src[1]) / T_src[2]
    elif rc in ("camera", "model"):
        d[0] = T_tgt[0] / T_tgt[2] - T_src[0] / T_src[2]
        d[1] = T_tgt[1] / T_tgt[2] - T_src[1] / T_src[2]
    else:
        raise Exception("Unknown: {}".format(rot_coord))
    d[2] = np.log(T_src[2] / T_tgt[2])
    return (d - T_means) / T_stds


def _axis_rot(axis, a):
    c, s = np.cos(a), np.sin(a)
    M = np.eye(3)
    i, j = [(1, 2), (2, 0), (0, 1)][axis]
    M[i, i], M[i, j], M[j, i], M[j, j] = c, -s, s, c
    return M


def euler2mat(ai, aj, ak):
    """RT_transform.py:250-317 with its default axes 'sxyz' (the only ones the EULER branches :139-140 / :39-40 use):
    rotations about the static x, y, z axes in that order = Rz(ak) Ry(aj) Rx(ai)."""
    return _axis_rot(2, ak) @ _axis_rot(1, aj) @ _axis_rot(0, ai)


def mat2euler(M):
    """RT_transform.py:320-383, axes 'sxyz' (gimbal lock: third angle := 0, :370-377)."""
    M = np.asarray(M, dtype=np.float64)[:3, :3]
    cy = np.hypot(M[0, 0], M[1, 0])
    if cy > np.finfo(float).eps * 4.0:
        return np.arctan2(M[2, 1], M[2, 2]), np.arctan2(-M[2, 0], cy), np.arctan2(M[1, 0], M[0, 0])
    return np.arctan2(-M[1, 2], M[1, 1]), np.arctan2(-M[2, 0], cy), 0.0


def RT_transform(pose_src, r, t, T_means, T_stds, rot_coord="MODEL"):
    """Compose a predicted (quat | euler, trans) delta onto pose_src. RT_transform.py:135-161."""
    r = np.squeeze(np.asarray(r, dtype=np.float64))
    if r.shape[0] == 3:
        Rm_delta = euler2mat(r[0], r[1], r[2])
    elif r.shape[0] == 4:
        Rm_delta = quat2mat(r / np.linalg.norm(r))
    else:
        raise Exception("Unknown r shape: {}".format(r.shape))
    t_delta = np.squeeze(t)
    if rot_coord.lower() == "naive":
        se3_mx = np.zeros((3, 4))
        se3_mx[:, :3] = Rm_delta
        se3_mx[:, 3] = t
        return se3_mul(se3_mx, pose_src)
    pose_est = np.zeros((3, 4))
    pose_est[:3, :3] = R_transform(pose_src[:3, :3], Rm_delta, rot_coord)
    pose_est[:3, 3] = T_transform(pose_src[:, 3], t_delta, T_means, T_stds, rot_coord)
    return pose_est


def calc_RT_delta(pose_src, pose_tgt, T_means, T_stds, rot_coord="MODEL", rot_type="MATRIX"):
    """Label generation (inverse of RT_transform). RT_transform.py:16-48."""
    if rot_coord.lower() == "naive":
        s2t = se3_mul(pose_tgt, se3_inverse(pose_src))
        Rm_delta, T_delta = s2t[:, :3], s2t[:, 3].reshape(3)
    else:
        Rm_delta = R_inv_transform(pose_src[:3, :3], pose_tgt[:3, :3], rot_coord)
        T_delta = T_inv_transform(pose_src[:, 3], pose_tgt[:, 3], T_means, T_stds, rot_coord)
    rt = rot_type.lower()
    if rt == "quat":
        r = mat2quat(Rm_delta)
    elif rt == "euler":
        r = mat2euler(Rm_delta)
    elif rt == "matrix":
        r = Rm_delta
    else:
        raise Exception("Unknown rot_type: {}".format(rot_type))
    return r, np.squeeze(T_delta)


def calc_se3(pose_src, pose_tgt):
    """RT_transform.py:186-197."""
    s2t = se3_mul(pose_tgt, se3_inverse(pose_src))
    return s2t[:, :3], s2t[:, 3].reshape(3)


def calc_rt_dist_m(pose_src, pose_tgt):
    """Geodesic rotation error (deg) + translation error (m). RT_transform.py:172-183."""
    from scipy.linalg import logm

    R_src, T_src = pose_src[:, :3], pose_src[:, 3]
    R_tgt, T_tgt = pose_tgt[:, :3], pose_tgt[:, 3]
    temp = logm(np.dot(np.transpose(R_src), R_tgt))
    rd_deg = np.linalg.norm(temp, "fro") / np.sqrt(2) / np.pi * 180
    return rd_deg, np.linalg.norm(T_tgt - T_src)
