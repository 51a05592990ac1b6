"""Transform3D custom op, numpy restatement (oracle; test-only).
Follows /root/reference/deepim/operator_py/transform3d.py: forward :42-118, backward :120-191,
T_transform_backward :193-225, quat2mat_forward :227-254, quat2mat_backward :256-327.
The reference's own self-check (:365-638) pins forward against RT_transform (1e-4) and backward against
finite differences (5e-3); tests re-create both."""
import numpy as np

from . import se3 as ose3

f32 = np.float32


def quat2mat_forward(q):
    w, x, y, z = np.asarray(q, dtype=f32)
    Nq = w * w + x * x + y * y + z * z
    if not (-1e-2 < Nq - 1 < 1e-2):
        return np.eye(3, dtype=f32)
    s = 2.0 / Nq
    X, Y, Z = x * s, y * s, z * s
    wX, wY, wZ = w * X, w * Y, w * Z
    xX, xY, xZ = x * X, x * Y, x * Z
    yY, yZ, zZ = y * Y, y * Z, z * Z
    return np.array([[1.0 - (yY + zZ), xY - wZ, xZ + wY], [xY + wZ, 1.0 - (xX + zZ), yZ - wX], [xZ - wY, yZ + wX, 1.0 - (xX + yY)]],
                    dtype=f32)


def forward(points, rotation, translation, pose_src, T_means, T_stds, rot_coord="CAMERA"):
    points = np.asarray(points, dtype=f32)
    B = points.shape[0]
    P = points.reshape(B, 3, -1)
    out = np.zeros_like(P)
    for b in range(B):
        Rd = quat2mat_forward(rotation[b])
        Rs = np.asarray(pose_src[b][:, :3], dtype=f32)
        Ts = np.asarray(pose_src[b][:, 3], dtype=f32)
        Rt = ose3.R_transform(Rs, Rd, rot_coord).astype(f32)
        if rot_coord.lower() == "naive":
            Tt = (np.dot(Rd, Ts.reshape(3, 1)) + np.asarray(translation[b], dtype=f32).reshape(3, 1)).reshape(3)
        else:
            Tt = ose3.T_transform(Ts, np.asarray(translation[b], dtype=f32), np.asarray(T_means, f32), np.asarray(T_stds, f32), rot_coord)
        out[b] = np.dot(Rt, P[b]) + np.asarray(Tt, dtype=f32).reshape(3, 1)
    return out.reshape(points.shape)


def backward(out_grad, points, rotation, translation, pose_src, T_means, T_stds, rot_coord="CAMERA"):
    points = np.asarray(points, dtype=f32)
    B = points.shape[0]
    P = points.reshape(B, 3, -1)
    G = np.asarray(out_grad, dtype=f32).reshape(B, 3, -1)
    T_means = np.asarray(T_means, f32)
    T_stds = np.asarray(T_stds, f32)
    d_rot = np.zeros((B, 4), dtype=f32)
    d_trans = np.zeros((B, 3), dtype=f32)
    rc = rot_coord.lower()
    for b in range(B):
        Rs = np.asarray(pose_src[b][:, :3], dtype=f32)
        Ts = np.asarray(pose_src[b][:, 3], dtype=f32)
        D = G[b].sum(axis=1)
        td = np.asarray(translation[b], dtype=f32)
        if rc == "naive":
            d_trans[b] = D
        else:
            t1 = td * T_stds + T_means
            z2 = Ts[2] / np.exp(t1[2])
            if rc in ("camera", "model"):
                share = -T_stds[2] * z2
                d_trans[b, 0] = D[0] * (T_stds[0] * z2)
                d_trans[b, 1] = D[1] * (T_stds[1] * z2)
                d_trans[b, 2] = D[0] * (share * (t1[0] + Ts[0] / Ts[2])) + D[1] * (share * (t1[1] + Ts[1] / Ts[2])) + D[2] * (-T_stds[2] * z2)
            else:
                d_trans[b, 0] = D[0] * (T_stds[0] * Ts[2])
                d_trans[b, 1] = D[1] * (T_stds[1] * Ts[2])
                d_trans[b, 2] = D[2] * (-T_stds[2] * z2)
        Rt_diff = np.dot(G[b], P[b].T)
        if rc == "model":
            Rd_diff = np.dot(Rs.T, Rt_diff)
        elif rc in ("camera", "camera_new"):
            Rd_diff = np.dot(Rt_diff, Rs.T)
        else:
            src = np.dot(Rs, P[b]) + Ts.reshape(3, 1)
            Rd_diff = np.dot(G[b], src.T)
        d_rot[b] = quat2mat_backward(Rd_diff, rotation[b])
    return d_rot, d_trans


def quat2mat_backward(D, q):
    w, x, y, z = np.asarray(q, dtype=f32)
    Nq = w * w + x * x + y * y + z * z
    Ns = np.sqrt(Nq)
    w_, x_, y_, z_ = np.asarray(q, dtype=f32) / Ns
    s = 2.0
    if not (-1e-4 < Nq - 1 < 1e-4):
        return np.zeros(4, dtype=f32)
    wd = (-z_ * D[0, 1] + y_ * D[0, 2] + z_ * D[1, 0] - x_ * D[1, 2] - y_ * D[2, 0] + x_ * D[2, 1]) * s
    xd = (y_ * D[0, 1] + z_ * D[0, 2] + y_ * D[1, 0] - 2 * x_ * D[1, 1] - w_ * D[1, 2] + z_ * D[2, 0] + w_ * D[2, 1] - 2 * x_ * D[2, 2]) * s
    yd = (-2 * y_ * D[0, 0] + x_ * D[0, 1] + w_ * D[0, 2] + x_ * D[1, 0] + z_ * D[1, 2] - w_ * D[2, 0] + z_ * D[2, 1] - 2 * y_ * D[2, 2]) * s
    zd = (-2 * z_ * D[0, 0] - w_ * D[0, 1] + x_ * D[0, 2] + w_ * D[1, 0] - 2 * z_ * D[1, 1] + y_ * D[1, 2] + x_ * D[2, 0] + y_ * D[2, 1]) * s
    share = Ns ** 3 * (w * wd + x * xd + y * yd + z * zd)
    return np.array([Ns * wd - w * share, Ns * xd - x * share, Ns * yd - y * share, Ns * zd - z * share], dtype=f32)
