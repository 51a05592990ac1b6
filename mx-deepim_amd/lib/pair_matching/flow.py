"""Dense flow labels from a rendered depth map and the two poses -- the numpy path the data loader uses for the FIRST iteration's
labels (reference lib/pair_matching/flow.py:12-81; the later iterations are re-labelled on the device by dim_depth_to_flow, the
restatement of lib/flow_c/gpu_flow_kernel.cu, whose visibility predicate differs slightly: SURVEY.md 8c).

    X   = depth_src * K^-1 [u, v, 1]                      back-projection of every source pixel
    Xp  = K (pose_tgt o pose_src^-1) X                    se3_mul / se3_inverse round to float32, like the reference
    (pw, ph) = Xp.xy / (Xp.z + 1e-15)                      target pixel
    visible  = depth_src != 0  and  round(pw, ph) inside the image  and  |depth_tgt[round] - Xp.z| < thresh  and  |depth_tgt[round]| > 1e-10
    flow     = (ph - v, pw - u)   ["[h, w]" order, standard_rep False]  or (pw - u, ph - v), zero where not visible
"""
import numpy as np

from lib.utils.projection import backproject_camera, se3_inverse, se3_mul


def calc_flow(depth_src, pose_src, pose_tgt, K, depth_tgt, thresh=3e-3, standard_rep=False):
    """-> flow (H,W,2), visible (H,W) in {0,1}, X_valid (3, n_visible) source points of the visible pixels"""
    depth_src = np.asarray(depth_src)
    H, W = depth_src.shape[:2]
    X = backproject_camera(depth_src, intrinsic_matrix=K)
    P = np.matmul(K, se3_mul(pose_tgt, se3_inverse(pose_src)))
    Xp = np.matmul(P, np.append(X, np.ones([1, X.shape[1]], dtype=np.float32), axis=0))
    pz = Xp[2] + 1e-15
    pw, ph = Xp[0] / pz, Xp[1] / pz

    src = np.flatnonzero(depth_src.ravel() != 0)
    col, row = np.round(pw[src]).astype(int), np.round(ph[src]).astype(int)
    inside = (col >= 0) & (col < W) & (row >= 0) & (row < H)
    d_hit = np.asarray(depth_tgt)[np.clip(row, 0, H - 1), np.clip(col, 0, W - 1)]
    seen = inside & (np.abs(d_hit - pz[src]) < thresh) & (np.abs(d_hit) > 1e-10)
    visible = np.zeros(H * W)
    visible[src[seen]] = 1
    visible = visible.reshape(H, W)

    u, v = np.meshgrid(np.linspace(0, W - 1, W), np.linspace(0, H - 1, H))
    du, dv = pw.reshape(H, W) - u, ph.reshape(H, W) - v
    flow = np.dstack([du, dv] if standard_rep else [dv, du])
    flow[visible != 1] = 0
    assert np.isnan(flow).sum() == 0
    X_valid = X[:, visible.ravel() != 0]
    return flow, visible, X_valid
