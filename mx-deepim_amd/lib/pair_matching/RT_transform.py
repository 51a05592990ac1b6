"""SE(3) pose parameterisation of DeepIM on the HIP kernels.

Same function names as /root/reference/lib/pair_matching/RT_transform.py (`RT_transform` :135-161, `calc_RT_delta` :16-48)
for single 3x4 numpy poses, plus the batched device forms the refinement loop and the training updater use
(`RT_transform_batch`, `calc_RT_delta_batch`).  Arithmetic is float64 inside the kernel (csrc/se3.hip), inputs/outputs float32.
"""
import numpy as np
import torch

from lib.hip import ops


def RT_transform_batch(pose_src, se3, T_means, T_stds, rot_coord="MODEL", out=None, out_f64=None):
    """pose_src (B,3,4), se3 (B,7) = [quat (raw), trans] CUDA f32 -> (B,3,4)."""
    return ops.se3_compose(pose_src, se3, rot_coord, T_means, T_stds, out=out, out_f64=out_f64)


def calc_RT_delta_batch(pose_src, pose_tgt, T_means, T_stds, rot_coord="MODEL"):
    """-> (quat (B,4) with w >= 0, trans (B,3)); rot_type 'QUAT' of the reference."""
    return ops.se3_delta(pose_src, pose_tgt, rot_coord, T_means, T_stds)


def _dev(a, device):
    return torch.as_tensor(np.ascontiguousarray(np.asarray(a, dtype=np.float32))).to(device)


def RT_transform(pose_src, r, t, T_means, T_stds, rot_coord="MODEL", device="cuda:0"):
    r = np.squeeze(r)
    if r.shape[0] not in (3, 4):   # 4 numbers: quaternion, 3: Euler angles (reference :138-146)
        raise Exception("Unknown r shape: {}".format(r.shape))
    se3 = np.concatenate([r, np.squeeze(t)]).reshape(1, r.shape[0] + 3)
    out64 = torch.empty((1, 3, 4), dtype=torch.float64, device=device)
    compose = ops.se3_compose if r.shape[0] == 4 else ops.se3_compose_euler
    compose(_dev(np.asarray(pose_src).reshape(1, 3, 4), device), _dev(se3, device), rot_coord, T_means, T_stds, out_f64=out64)
    return out64[0].cpu().numpy()


def calc_RT_delta(pose_src, pose_tgt, T_means, T_stds, rot_coord="MODEL", rot_type="MATRIX", device="cuda:0"):
    """reference :16-48.  rot_type "quat" -> (w,x,y,z) with w >= 0, "matrix" -> the 3x3 residual rotation (the default there),
    "euler" -> the three static-xyz angles of mat2euler (:320-383)."""
    kind = rot_type.lower()
    ps, pt = _dev(np.asarray(pose_src).reshape(1, 3, 4), device), _dev(np.asarray(pose_tgt).reshape(1, 3, 4), device)
    if kind == "quat":
        r, t = calc_RT_delta_batch(ps, pt, T_means, T_stds, rot_coord)
    elif kind == "matrix":
        r, t = ops.se3_delta_matrix(ps, pt, rot_coord, T_means, T_stds)
    elif kind == "euler":
        r, t = ops.se3_delta_euler(ps, pt, rot_coord, T_means, T_stds)
    else:
        raise Exception("Unknown rot_type: {}".format(rot_type))
    return r[0].cpu().numpy(), t[0].cpu().numpy()
