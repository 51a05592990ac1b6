"""Zoom custom-ops of DeepIM, numpy restatement (oracle; test-only).

Restates deepim/operator_py/zoom_mask.py, zoom_image_with_factor.py,
zoom_image.py, zoom_mask_with_factor.py, zoom_flow.py, zoom_depth.py,
zoom_trans.py of the reference.  The two MXNet primitives they call
(GridGenerator(affine) and BilinearSampler) are NOT in /root/reference
(module mxnet ~1.2.0, absent) and are restated from their published
semantics: "parity unpinned" for those two, see SURVEY.md 8(a) notes.

All sampling arithmetic is float32 like MXNet's; bbox / zoom-factor
arithmetic is float64 numpy like the reference host code, then cast to f32.
"""
import numpy as np

f32 = np.float32


def mx_round(x):
    """mx.nd.round: half away from zero."""
    return np.sign(x) * np.floor(np.abs(x) + f32(0.5))


def affine_grid(theta, H, W):
    """mx.nd.GridGenerator(transform_type='affine').
    theta (B,6) rows [a,b,c,d,e,f]; out (B,2,H,W): x_s = a*x_t+b*y_t+c, y_s = d*x_t+e*y_t+f
    with x_t = -1 + 2j/(W-1), y_t = -1 + 2i/(H-1)."""
    theta = np.asarray(theta, dtype=f32).reshape(-1, 6)
    xt = (f32(-1.0) + np.arange(W, dtype=f32) * f32(2.0 / (W - 1))).astype(f32)
    yt = (f32(-1.0) + np.arange(H, dtype=f32) * f32(2.0 / (H - 1))).astype(f32)
    B = theta.shape[0]
    grid = np.empty((B, 2, H, W), dtype=f32)
    for b in range(B):
        a, bb, c, d, e, ff = theta[b]
        grid[b, 0] = (a * xt[None, :] + bb * yt[:, None]) + c
        grid[b, 1] = (d * xt[None, :] + e * yt[:, None]) + ff
    return grid


def bilinear_sampler(data, grid):
    """mx.nd.BilinearSampler: x_real=(gx+1)(W-1)/2, zero for out-of-range corners."""
    data = np.asarray(data, dtype=f32)
    B, C, H, W = data.shape
    gx = grid[:, 0]
    gy = grid[:, 1]
    x_real = (gx + f32(1)) * f32(W - 1) / f32(2)
    y_real = (gy + f32(1)) * f32(H - 1) / f32(2)
    x0f = np.floor(x_real)
    y0f = np.floor(y_real)
    wx0 = (f32(1.0) - (x_real - x0f)).astype(f32)  # top_left_x_w
    wy0 = (f32(1.0) - (y_real - y0f)).astype(f32)
    # clip before int-cast so absurd coordinates cannot overflow; such corners are invalid anyway
    x0 = np.clip(x0f, -2, W + 1).astype(np.int64)
    y0 = np.clip(y0f, -2, H + 1).astype(np.int64)
    out = np.zeros((B, C) + gx.shape[1:], dtype=f32)

    def corner(yy, xx):
        valid = (xx >= 0) & (xx <= W - 1) & (yy >= 0) & (yy <= H - 1)
        xc = np.clip(xx, 0, W - 1)
        yc = np.clip(yy, 0, H - 1)
        v = np.empty((B, C) + xx.shape[1:], dtype=f32)
        for b in range(B):
            v[b] = data[b][:, yc[b], xc[b]]
        return v * valid[:, None].astype(f32)

    tl = corner(y0, x0)
    tr = corner(y0, x0 + 1)
    bl = corner(y0 + 1, x0)
    br = corner(y0 + 1, x0 + 1)
    wx0 = wx0[:, None]
    wy0 = wy0[:, None]
    one = f32(1.0)
    out = tl * wy0 * wx0 + tr * wy0 * (one - wx0) + bl * (one - wy0) * wx0 + br * (one - wy0) * (one - wx0)
    return out.astype(f32)


def _grid_from_factor(zoom_factor, H, W):
    zf = np.asarray(zoom_factor, dtype=f32).reshape(-1, 4)
    theta = np.zeros((zf.shape[0], 6), dtype=f32)
    theta[:, 0] = zf[:, 0]
    theta[:, 2] = zf[:, 2]
    theta[:, 4] = zf[:, 1]
    theta[:, 5] = zf[:, 3]
    return affine_grid(theta, H, W)


def inverse_zoom_factor(zoom_factor, H, W):
    """zoom_flow.py:35-44 / zoom_mask_with_factor.py:44-53 (float32 numpy scalars there)."""
    zf = np.asarray(zoom_factor, dtype=f32).reshape(-1, 4)
    out = np.zeros_like(zf)
    for b in range(zf.shape[0]):
        wx_in, wy_in, tx_in, ty_in = zf[b]
        wx = 1 / wx_in
        wy = 1 / wy_in
        crop_w = wx_in * W
        crop_h = wy_in * H
        cx = tx_in * 0.5 * W + 0.5 * W
        cy = ty_in * 0.5 * H + 0.5 * H
        tx = (W * 0.5 - cx) / crop_w * 2
        ty = (H * 0.5 - cy) / crop_h * 2
        out[b] = [wx, wy, tx, ty]
    return out.astype(f32)


def _bbox(valid):
    """min/max of non-zero columns/rows. zoom_mask.py:55-62."""
    nz_x = np.nonzero(np.max(valid, axis=0))[0]
    nz_y = np.nonzero(np.max(valid, axis=1))[0]
    return nz_x, nz_y


def zoom_factor_from_valid(valid_real, valid_rendered, src_pose, K, H, W):
    """zoom_mask.py:50-117 / zoom_image.py:41-100 per-sample zoom-window rule.
    valid_*: (B,H,W) bool; src_pose (B,3,4) f32; K (3,3) f32. Returns (B,4) f32, and a
    per-sample flag 'rendered_empty'."""
    B = valid_real.shape[0]
    K = np.asarray(K, dtype=f32)
    src_pose = np.asarray(src_pose, dtype=f32)
    zf = np.zeros((B, 4), dtype=f32)
    empty = np.zeros((B,), dtype=bool)
    for b in range(B):
        nz_x, nz_y = _bbox(valid_real[b])
        rsx, rex, rsy, rey = np.min(nz_x), np.max(nz_x), np.min(nz_y), np.max(nz_y)
        rcx = (rsx + rex) * 0.5
        rcy = (rsy + rey) * 0.5
        nz_x, nz_y = _bbox(valid_rendered[b])
        c = np.dot(K, src_pose[b][:, 3])  # float32 dot
        ccx = c[0] / c[2]
        ccy = c[1] / c[2]
        if len(nz_x) == 0 or len(nz_y) == 0:
            empty[b] = True
            dsx, dex, dsy, dey = rsx, rex, rsy, rey
            zcx, zcy = rcx, rcy
        else:
            dsx, dex, dsy, dey = np.min(nz_x), np.max(nz_x), np.min(nz_y), np.max(nz_y)
            zcx, zcy = ccx, ccy
        left = max(zcx - dsx, zcx - rsx)
        right = max(dex - zcx, rex - zcx)
        up = max(zcy - dsy, zcy - rsy)
        down = max(rey - zcy, dey - zcy)
        crop_h = np.max([0.75 * right, 0.75 * left, up, down]) * 1.4 * 2
        wx = crop_h / H
        tx = zcx / W * 2 - 1
        ty = zcy / H * 2 - 1
        zf[b] = [wx, wx, tx, ty]
    return zf, empty


def zoom_mask(mask_observed, mask_gt_observed, mask_rendered, src_pose, K, H=480, W=640):
    """ZoomMaskOperator.forward, zoom_mask.py:29-134.
    Returns zoom_mask_observed, zoom_mask_gt_observed, zoom_mask_rendered, zoom_factor."""
    mo = np.asarray(mask_observed, dtype=f32)
    mg = np.asarray(mask_gt_observed, dtype=f32)
    mr = np.asarray(mask_rendered, dtype=f32).copy()
    valid_real = np.sum(mg, axis=1) > 0.3
    mr_bin = np.where(mr > 0.2, f32(1), f32(0)).astype(f32)
    valid_rend = np.sum(mr_bin, axis=1) > 0.3
    zf, _ = zoom_factor_from_valid(valid_real, valid_rend, src_pose, K, H, W)
    grid = _grid_from_factor(zf, H, W)
    return (
        mx_round(bilinear_sampler(mo, grid)),
        mx_round(bilinear_sampler(mg, grid)),
        mx_round(bilinear_sampler(mr_bin, grid)),
        zf,
    )


def zoom_image_with_factor(zoom_factor, image_observed, image_rendered, pixel_means, H=480, W=640):
    """ZoomImageWithFactorOperator.forward, zoom_image_with_factor.py:31-75.
    pixel_means = config.network.PIXEL_MEANS (the Prop reverses it, :94)."""
    pm = np.asarray(pixel_means, dtype=f32).reshape(3)[::-1].reshape(1, 3, 1, 1)
    grid = _grid_from_factor(zoom_factor, H, W)
    io = bilinear_sampler(np.asarray(image_observed, dtype=f32) + pm, grid) - pm
    ir = bilinear_sampler(np.asarray(image_rendered, dtype=f32) + pm, grid) - pm
    return io.astype(f32), ir.astype(f32)


def zoom_image(image_observed, image_rendered, src_pose, K, pixel_means, H=480, W=640):
    """ZoomImageOperator.forward, zoom_image.py:26-119 (no-mask configs)."""
    pm = np.asarray(pixel_means, dtype=f32).reshape(3)[::-1].reshape(1, 3, 1, 1)
    io = np.asarray(image_observed, dtype=f32) + pm
    ir = np.asarray(image_rendered, dtype=f32) + pm
    valid_real = np.sum(io, axis=1) > 0.01
    valid_rend = np.sum(ir, axis=1) > 0.01
    zf, _ = zoom_factor_from_valid(valid_real, valid_rend, src_pose, K, H, W)
    grid = _grid_from_factor(zf, H, W)
    return (bilinear_sampler(io, grid) - pm).astype(f32), (bilinear_sampler(ir, grid) - pm).astype(f32), zf


def zoom_mask_with_factor(zoom_factor, mask, b_inv_zoom=False, H=480, W=640):
    """ZoomMaskWithFactorOperator.forward, zoom_mask_with_factor.py:29-68."""
    m = np.asarray(mask, dtype=f32)
    m = np.where(m > 0.2, f32(1), f32(0)).astype(f32)
    zf = inverse_zoom_factor(zoom_factor, H, W) if b_inv_zoom else np.asarray(zoom_factor, dtype=f32)
    return mx_round(bilinear_sampler(m, _grid_from_factor(zf, H, W)))


def zoom_flow(zoom_factor, flow, flow_weights=None, b_inv_zoom=False, H=480, W=640):
    """ZoomFlowOperator.forward, zoom_flow.py:28-77."""
    zf_in = np.asarray(zoom_factor, dtype=f32).reshape(-1, 4)
    zf = inverse_zoom_factor(zf_in, H, W) if b_inv_zoom else zf_in
    grid = _grid_from_factor(zf, H, W)
    zflow = bilinear_sampler(flow, grid)
    wx = zf_in[:, 0].reshape(-1, 1, 1, 1)
    zflow = (zflow * wx if b_inv_zoom else zflow / wx).astype(f32)
    if b_inv_zoom:
        return zflow
    zw = mx_round(bilinear_sampler(flow_weights, grid) - f32(0.45))
    return zflow, zw.astype(f32)


def zoom_depth(zoom_factor, depth_observed, depth_rendered, H=480, W=640):
    """ZoomDepthOperator.forward, zoom_depth.py:24-50."""
    grid = _grid_from_factor(zoom_factor, H, W)
    return bilinear_sampler(depth_observed, grid), bilinear_sampler(depth_rendered, grid)


def zoom_trans(zoom_factor, trans_delta, b_inv_zoom=False):
    """ZoomTransOperator.forward, zoom_trans.py:22-53 (float64 accumulate, f32 out)."""
    zf = np.asarray(zoom_factor, dtype=f32).reshape(-1, 4)
    td = np.asarray(trans_delta, dtype=f32)
    out = np.zeros(td.shape)
    for b in range(td.shape[0]):
        wx = zf[b][0]
        dx, dy, dz = td[b]
        if b_inv_zoom:
            out[b] = [dx * wx, dy * wx, dz]
        else:
            out[b] = [dx / wx, dy / wx, dz]
    return out.astype(f32)


def zoom_trans_backward(zoom_factor, out_grad, b_inv_zoom=False, b_zoom_grad=False):
    """ZoomTransOperator.backward, zoom_trans.py:55-76."""
    zf = np.asarray(zoom_factor, dtype=f32).reshape(-1, 4)
    g = np.asarray(out_grad, dtype=f32)
    out = np.zeros(g.shape)
    for b in range(g.shape[0]):
        wx = zf[b][0]
        gx, gy, gz = g[b]
        if b_zoom_grad:
            if b_inv_zoom:
                gx, gy = gx * wx, gy * wx
            else:
                gx, gy = gx / wx, gy / wx
        out[b] = [gx, gy, gz]
    return out.astype(f32)
