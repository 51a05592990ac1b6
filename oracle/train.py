"""Training graph + gradients + SGD, torch-CPU autograd restatement (oracle; test-only).

Restates /root/reference/deepim/symbols/deepIM_flownet.py get_train_symbol :562-762 and get_loss :303-560 for the shipped
configuration (INPUT_MASK, PRED_MASK, PRED_FLOW, SE3_PM_LOSS type L1, no SE3_DIST_LOSS) and its variants (first-layer input without
masks / with depth planes :33-66, SE3_DIST_LOSS, the other point-matching / translation loss types), with the MXNet-internal gradient
conventions listed in SURVEY.md A11 ("parity unpinned": MXNet is absent):
  MakeLoss backward = grad_scale * dloss/dx; LogisticRegressionOutput backward = grad_scale/num_output * (sigmoid(x) - y);
  gradients summed over the batch; Transform3D / ZoomTrans use the reference's HAND-WRITTEN backward (oracle.transform3d,
  zoom_trans.py:55-76), not autograd.
"""
import numpy as np
import torch
import torch.nn.functional as F

from . import flownet as oflow, transform3d as ot3d, zoom as ozoom


class _Transform3D(torch.autograd.Function):
    @staticmethod
    def forward(ctx, points, rot, trans, pose_src, T_means, T_stds, rot_coord):
        ctx.save = (points.detach().numpy(), rot.detach().numpy().astype(np.float32), trans.detach().numpy().astype(np.float32), pose_src,
                    T_means, T_stds, rot_coord)
        out = ot3d.forward(ctx.save[0], ctx.save[1], ctx.save[2], pose_src, T_means, T_stds, rot_coord)
        return torch.from_numpy(out.astype(np.float64))

    @staticmethod
    def backward(ctx, g):
        pts, rot, trans, pose_src, T_means, T_stds, rot_coord = ctx.save
        d_rot, d_trans = ot3d.backward(g.numpy().astype(np.float32), pts, rot, trans, pose_src, T_means, T_stds, rot_coord)
        return None, torch.from_numpy(d_rot.astype(np.float64)), torch.from_numpy(d_trans.astype(np.float64)), None, None, None, None


class _InvZoomTrans(torch.autograd.Function):
    """ZoomTrans(b_inv_zoom=True, b_zoom_grad=False): forward (dx,dy)*wx, backward identity (zoom_trans.py:37-41, :64-72)."""

    @staticmethod
    def forward(ctx, t, wx):
        out = t.clone()
        out[:, :2] = t[:, :2] * wx[:, None]
        return out

    @staticmethod
    def backward(ctx, g):
        return g, None


def _elem_loss(x, kind, scalar):
    """|x|, x^2, or mx.sym.smooth_l1(x, scalar): 0.5 (scalar x)^2 where |x| < 1 / scalar^2, |x| - 0.5 / scalar^2 elsewhere"""
    if kind == "L1":
        return x.abs()
    if kind == "L2":
        return x * x
    if kind == "smooth_L1":
        s2 = float(scalar) ** 2
        return torch.where(x.abs() < 1.0 / s2, 0.5 * s2 * x * x, x.abs() - 0.5 / s2)
    raise Exception("Unknown loss type: {}".format(kind))


def loss_and_grads(params, batch, cfg, K, dtype=torch.float64):
    """-> (outputs dict, grads dict name -> numpy in MXNet layouts).  batch: numpy blobs with the reference names."""
    H, W = 480, 640
    ti = cfg.train_iter
    pred_flow, pred_mask = bool(cfg.network.PRED_FLOW), bool(cfg.network.PRED_MASK)
    zmo = zmg = zmr = None
    if cfg.network.INPUT_MASK or pred_mask:   # :589-623 ZoomMask + ZoomImageWithFactor; else :625-640 ZoomImage
        zmo, zmg, zmr, zf = ozoom.zoom_mask(batch["mask_observed"], batch["mask_gt_observed"], batch["mask_rendered"], batch["src_pose"], K, H, W)
        zio, zir = ozoom.zoom_image_with_factor(zf, batch["image_observed"], batch["image_rendered"], cfg.network.PIXEL_MEANS, H, W)
    else:
        zio, zir, zf = ozoom.zoom_image(batch["image_observed"], batch["image_rendered"], batch["src_pose"], K, cfg.network.PIXEL_MEANS, H, W)
    if pred_flow:
        zflow, zfw = ozoom.zoom_flow(zf, batch["flow"], batch["flow_weights"], b_inv_zoom=False, H=H, W=W)
    # the Concat of get_convs (deepIM_flownet.py:33-66) for this configuration: the zoom window always comes from the masks in
    # training (:589-612, PRED_MASK), the masks are network inputs only with INPUT_MASK, the zoomed depth planes with INPUT_DEPTH (:670-683)
    with_masks = bool(cfg.network.INPUT_MASK and cfg.network.PRED_MASK)
    zdo = zdr = None
    if cfg.network.INPUT_DEPTH:
        zdo, zdr = ozoom.zoom_depth(zf, batch["depth_observed"], batch["depth_rendered"], H, W)
    data = oflow.network_input(zio, zir, zmo if with_masks else None, zmr if with_masks else None, zdo, zdr)
    P = {k: torch.from_numpy(np.ascontiguousarray(v)).to(dtype).requires_grad_(True) for k, v in params.items()}
    x = torch.from_numpy(data).to(dtype)
    feats = {}
    from .flownet import ENCODER, crop_like

    for name, cout, k, s, p in ENCODER:
        x = F.leaky_relu(F.conv2d(x, P[name + "_weight"], P[name + "_bias"], stride=s, padding=p), 0.1)
        feats[name] = x
    fc6 = F.leaky_relu(F.linear(x.reshape(x.shape[0], -1), P["fc6_weight"], P["fc6_bias"]), 0.1)
    fc7 = F.leaky_relu(F.linear(fc6, P["fc7_weight"], P["fc7_bias"]), 0.1)
    rot = F.linear(fc7, P["rot_weight"], P["rot_bias"])
    tz = F.linear(fc7, P["trans_weight"], P["trans_bias"])
    rot_norm = rot / torch.sqrt((rot * rot).sum(dim=1, keepdim=True) + 1e-10)  # L2Normalization(instance)
    trans_est = _InvZoomTrans.apply(tz, torch.from_numpy(zf[:, 0].astype(np.float64)).to(dtype))
    # decoder (:213-299: built when either head is predicted) and the heads that exist
    r10, r8, r6 = feats["conv6_1"], feats["conv5_1"], feats["conv4_1"]
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dtype)  # noqa: E731
    cat2 = cat3 = flow_est = logit = None
    flow_loss_ = torch.zeros(1, dtype=dtype)
    L = torch.zeros((), dtype=dtype)
    if pred_flow or pred_mask:
        c1 = F.conv2d(r10, P["Convolution1_weight"], P["Convolution1_bias"], padding=1)
        d5 = F.leaky_relu(crop_like(F.conv_transpose2d(r10, P["deconv5_weight"], P["deconv5_bias"], stride=2), r8.shape[2:], (1, 1)), 0.1)
        u65 = crop_like(F.conv_transpose2d(c1, P["upsample_flow6to5_weight"], P["upsample_flow6to5_bias"], stride=2), r8.shape[2:], (1, 1))
        cat2 = torch.cat([r8, d5, u65], dim=1)
        c2 = F.conv2d(cat2, P["Convolution2_weight"], P["Convolution2_bias"], padding=1)
        d4 = F.leaky_relu(crop_like(F.conv_transpose2d(cat2, P["deconv4_weight"], P["deconv4_bias"], stride=2), r6.shape[2:], (1, 1)), 0.1)
        u54 = crop_like(F.conv_transpose2d(c2, P["upsample_flow5to4_weight"], P["upsample_flow5to4_bias"], stride=2), r6.shape[2:], (1, 1))
        cat3 = torch.cat([r6, d4, u54], dim=1)
    if pred_flow:
        f4 = F.conv2d(cat3, P["Convolution3_weight"], P["Convolution3_bias"], padding=1)
        flow_est = crop_like(F.conv_transpose2d(f4, P["upsampling_weight"], None, stride=16, groups=2), (H, W), (8, 8))
        flow_loss_ = t(zfw) * (flow_est - t(zflow) / cfg.dataset.NORMALIZE_FLOW) ** 2
        L = L + (ti.LW_FLOW / (480.0 * 640.0)) * flow_loss_.sum()
    if pred_mask:
        m4 = F.conv2d(cat3, P["mask_conv3_weight"], P["mask_conv3_bias"], padding=1)
        logit = crop_like(F.conv_transpose2d(m4, P["mask_upsampling_weight"], None, stride=16), (H, W), (8, 8))
        bce = F.binary_cross_entropy_with_logits(logit, t(zmg), reduction="sum")  # d/dx = sigmoid(x) - y
        L = L + (ti.LW_MASK / (480.0 * 640.0)) * bce
    # losses -> one scalar whose autograd gradient equals MXNet's head gradients
    pts_est = _Transform3D.apply(t(batch["point_cloud_model"]), rot_norm, trans_est, batch["src_pose"], np.asarray(cfg.dataset.trans_means, np.float32),
                                 np.asarray(cfg.dataset.trans_stds, np.float32), cfg.network.ROT_COORD)
    # point matching: SE3_PM_LOSS_TYPE 'L1' | 'L2' | 'smooth_L1' (deepIM_flownet.py:458-499)
    pm_loss_ = t(batch["point_cloud_weights"]) * _elem_loss((pts_est - t(batch["point_cloud_observed"])) / cfg.dataset.NORMALIZE_3D_POINT,
                                                             ti.SE3_PM_LOSS_TYPE, ti.SE3_PM_SL1_SCALAR)
    if ti.SE3_PM_LOSS:
        L = L + (ti.LW_PM / float(ti.NUM_3D_SAMPLE)) * pm_loss_.sum()
    rot_loss_ = trans_loss_ = torch.zeros(1, dtype=dtype)
    if ti.SE3_DIST_LOSS:
        # deepIM_flownet.py:396-437: rot_loss = 1 - (rot_gt . rot_est_norm)^2, trans_loss = TRANS_LOSS_TYPE(zoom_trans_est - zoom_trans_gt);
        # zoom_trans_gt = ZoomTrans(zoom_factor, trans label, b_inv_zoom False) = (dx, dy) / wx, dz (:659-665, zoom_trans.py:37-41)
        rot_loss_ = 1.0 - ((t(batch["rot"]) * rot_norm).sum(dim=1)) ** 2
        zt_gt = t(batch["trans"]).clone()
        zt_gt[:, :2] = zt_gt[:, :2] / torch.from_numpy(zf[:, 0].astype(np.float64)).to(dtype)[:, None]
        trans_loss_ = _elem_loss(tz - zt_gt, ti.TRANS_LOSS_TYPE, ti.TRANS_SMOOTH_L1_SCALAR)
        L = L + ti.LW_ROT * rot_loss_.sum() + ti.LW_TRANS * trans_loss_.sum()
    for tnsr in (cat2, cat3, r10, r8, r6):
        if tnsr is not None:
            tnsr.retain_grad()
    L.backward()
    grads = {k: (v.grad.numpy().astype(np.float64) if v.grad is not None else np.zeros(v.shape)) for k, v in P.items()}
    for k in ("upsampling_weight", "mask_upsampling_weight"):  # lr_mult 0: frozen
        if k in grads:
            grads[k] = np.zeros_like(grads[k])
    out = {"rot_est_norm": rot_norm.detach().numpy(), "trans_est": trans_est.detach().numpy(), "zoom_factor": zf, "r10": r10.detach().numpy(),
           "d_r10": r10.grad.numpy(), "d_r8": r8.grad.numpy(), "d_r6": r6.grad.numpy(), "flow_loss_sum": float(flow_loss_.sum()), "pm_loss_sum": float(pm_loss_.sum()),
           "rot_loss_sum": float(rot_loss_.sum()), "trans_loss_sum": float(trans_loss_.sum())}
    if flow_est is not None:
        out["flow_est_crop"] = flow_est.detach().numpy()
    if logit is not None:
        out["mask_logit"] = logit.detach().numpy()
    if cat2 is not None:
        out.update({"cat2": cat2.detach().numpy(), "cat3": cat3.detach().numpy(), "d_cat2": cat2.grad.numpy(), "d_cat3": cat3.grad.numpy()})
    return out, grads


def sgd_step(params, grads, moms, lr, momentum, wd):
    """mx.optimizer.SGD: mom = momentum*mom - lr*(g + wd*w); w += mom; wd_mult = 0 unless the name ends with _weight; frozen skipped."""
    for k in params:
        if k in ("upsampling_weight", "mask_upsampling_weight"):
            continue
        w = params[k].astype(np.float64)
        m = momentum * moms[k] - lr * (grads[k] + (wd if k.endswith("_weight") else 0.0) * w)
        moms[k] = m
        params[k] = (w + m).astype(np.float32)
    return params, moms


def adam_step(params, grads, means, variances, t, lr, beta1=0.9, beta2=0.999, epsilon=1e-8, wd=0.0, rescale_grad=1.0):
    """mx.optimizer.Adam.update + adam_update op (the reference selects it with TRAIN.optimizer == 'adam', deepim/train.py:338-375):
    lr_t = lr*sqrt(1-beta2^t)/(1-beta1^t); g' = rescale*g + wd*w; mean, var EMAs; w -= lr_t*mean/(sqrt(var)+eps).  float32 like the op."""
    lr_t = np.float32(lr * np.sqrt(1.0 - beta2 ** t) / (1.0 - beta1 ** t))
    b1, b2 = np.float32(beta1), np.float32(beta2)
    for k in params:
        if k in ("upsampling_weight", "mask_upsampling_weight"):
            continue
        w = params[k].astype(np.float32)
        g = np.float32(rescale_grad) * grads[k].astype(np.float32) + np.float32(wd) * w
        means[k] = b1 * means[k].astype(np.float32) + (np.float32(1) - b1) * g
        variances[k] = b2 * variances[k].astype(np.float32) + (np.float32(1) - b2) * g * g
        params[k] = (w - lr_t * means[k] / (np.sqrt(variances[k]) + np.float32(epsilon))).astype(np.float32)
    return params, means, variances
