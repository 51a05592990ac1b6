"""Thin torch-tensor front-ends of the C ABI (one function per entry point of include/deepim_hip.h).

Every function only enqueues HIP kernels on torch's current stream; outputs are written into
caller-provided tensors when given, so the refinement loop can run allocation-free inside a
hipGraph capture.  No CPU fallback: non-CUDA tensors raise DeepIMHipError.
"""
import struct

import numpy as np

import torch

from . import capi
from .capi import check, current_stream, dptr, host_f32, lib

f32 = torch.float32
i32 = torch.int32
bf16 = torch.bfloat16


def to_bf16(src, out=None):
    """element-wise f32 -> bf16 (round to nearest even) as a kernel launch: the packed weight arrays of the bf16 convolution
    entry points, the flat gradient bucket"""
    out = out if out is not None else torch.empty(src.shape, dtype=bf16, device=src.device)
    check(lib().dim_f32_to_bf16(dptr(src, f32), dptr(out, bf16), src.numel(), current_stream()))
    return out


def from_bf16(src, out=None):
    out = out if out is not None else torch.empty(src.shape, dtype=f32, device=src.device)
    check(lib().dim_bf16_to_f32(dptr(src, bf16), dptr(out, f32), src.numel(), current_stream()))
    return out


def _wp(w):
    """(device pointer, is_bf16) of a packed weight array: bf16 arrays select the bf16 matrix-pipe twin of the entry point"""
    if w.dtype == bf16:
        return dptr(w, bf16), True
    return dptr(w, f32), False


def _new(shape, like, dtype=f32):
    return torch.empty(shape, dtype=dtype, device=like.device)


def mask_bbox(x, thr, mode=0, means3=None, out=None):
    """bbox (B,4) int32 = {min_x,max_x,min_y,max_y} of the predicate; empty = {W,-1,H,-1}."""
    B, C, H, W = x.shape
    out = out if out is not None else _new((B, 4), x, i32)
    keep, mp = host_f32(means3, 3) if means3 is not None else (None, None)
    check(lib().dim_mask_bbox(dptr(x, f32), B, C, H, W, mode, float(thr), mp, dptr(out, i32), current_stream()))
    return out


def zoom_factor(bbox_obs, bbox_ren, src_pose, K, H, W, out=None, status=None):
    B = src_pose.shape[0]
    out = out if out is not None else _new((B, 4), src_pose)
    keep, kp = host_f32(K, 9)
    check(lib().dim_zoom_factor(dptr(bbox_obs, i32), dptr(bbox_ren, i32), dptr(src_pose, f32), kp, B, H, W, dptr(out, f32),
                                dptr(status, i32), current_stream()))
    return out


def zoom_planes(x, zf, inverse=False, pre=0, post=0, add3=None, scale_mode=0, out=None):
    B, C, H, W = x.shape
    out = out if out is not None else torch.empty_like(x)
    keep, ap = host_f32(add3, 3) if add3 is not None else (None, None)
    check(lib().dim_zoom_planes(dptr(x, f32), dptr(zf, f32), dptr(out, f32), B, C, H, W, int(inverse), pre, post, ap, scale_mode,
                                current_stream()))
    return out


def zoom_net_input(img_obs, img_ren, mask_obs, mask_ren, zf, plane_means3, X=None, nchw_out=None):
    """X (B,H,W,8) NHWC network input.  nchw_out: optional (z_img_obs, z_img_ren, z_mask_obs, z_mask_ren)."""
    B, _, H, W = img_obs.shape
    X = X if X is not None else _new((B, H, W, 8), img_obs)
    keep, mp = host_f32(plane_means3, 3)
    z = nchw_out if nchw_out is not None else (None, None, None, None)
    check(lib().dim_zoom_net_input(dptr(img_obs, f32), dptr(img_ren, f32), dptr(mask_obs, f32), dptr(mask_ren, f32), dptr(zf, f32),
                                   dptr(X, f32), B, H, W, mp, dptr(z[0], f32), dptr(z[1], f32), dptr(z[2], f32), dptr(z[3], f32),
                                   current_stream()))
    return X


def zoom_net_input_ex(img_obs, img_ren, extra_obs, extra_ren, zf, plane_means3, mode, X=None):
    """the other input arities of the first layer: mode 1 = images only, mode 2 = depth planes in place of the masks (see the header)"""
    B, _, H, W = img_obs.shape
    X = X if X is not None else _new((B, H, W, 8), img_obs)
    keep, mp = host_f32(plane_means3, 3)
    check(lib().dim_zoom_net_input_ex(dptr(img_obs, f32), dptr(img_ren, f32), dptr(extra_obs, f32), dptr(extra_ren, f32), dptr(zf, f32),
                                      dptr(X, f32), B, H, W, mp, int(mode), current_stream()))
    return X


def zoom_trans(zf, t, mode, out=None):
    B = t.shape[0]
    out = out if out is not None else torch.empty_like(t)
    check(lib().dim_zoom_trans(dptr(zf, f32), dptr(t, f32), dptr(out, f32), B, mode, current_stream()))
    return out


def se3_compose(pose_src, se3, rot_coord, T_means, T_stds, out=None, out_f64=None):
    B = pose_src.shape[0]
    out = out if out is not None else torch.empty_like(pose_src)
    k1, mp = host_f32(T_means, 3)
    k2, sp = host_f32(T_stds, 3)
    check(lib().dim_se3_compose(dptr(pose_src, f32), dptr(se3, f32), dptr(out, f32), dptr(out_f64, torch.float64), B,
                                capi.rot_coord_id(rot_coord), mp, sp, current_stream()))
    return out


def se3_delta(pose_src, pose_tgt, rot_coord, T_means, T_stds):
    B = pose_src.shape[0]
    rot = _new((B, 4), pose_src)
    trans = _new((B, 3), pose_src)
    k1, mp = host_f32(T_means, 3)
    k2, sp = host_f32(T_stds, 3)
    check(lib().dim_se3_delta(dptr(pose_src, f32), dptr(pose_tgt, f32), dptr(rot, f32), dptr(trans, f32), B,
                              capi.rot_coord_id(rot_coord), mp, sp, current_stream()))
    return rot, trans


def se3_delta_matrix(pose_src, pose_tgt, rot_coord, T_means, T_stds):
    """-> (rotation residual (B,3,3), translation residual (B,3)): calc_RT_delta(..., rot_type="MATRIX")"""
    B = pose_src.shape[0]
    rot = _new((B, 3, 3), pose_src)
    trans = _new((B, 3), pose_src)
    k1, mp = host_f32(T_means, 3)
    k2, sp = host_f32(T_stds, 3)
    check(lib().dim_se3_delta_matrix(dptr(pose_src, f32), dptr(pose_tgt, f32), dptr(rot, f32), dptr(trans, f32), B,
                                     capi.rot_coord_id(rot_coord), mp, sp, current_stream()))
    return rot, trans


def se3_compose_euler(pose_src, euler_trans6, rot_coord, T_means, T_stds, out=None, out_f64=None):
    """RT_transform with a 3-number EULER rotation delta: euler_trans6 (B,6) = [ai, aj, ak, t]"""
    B = pose_src.shape[0]
    out = out if out is not None else torch.empty_like(pose_src)
    k1, mp = host_f32(T_means, 3)
    k2, sp = host_f32(T_stds, 3)
    check(lib().dim_se3_compose_euler(dptr(pose_src, f32), dptr(euler_trans6, f32), dptr(out, f32), dptr(out_f64, torch.float64), B,
                                      capi.rot_coord_id(rot_coord), mp, sp, current_stream()))
    return out


def se3_delta_euler(pose_src, pose_tgt, rot_coord, T_means, T_stds):
    """-> (static-xyz Euler angles (B,3), translation residual (B,3)): calc_RT_delta(..., rot_type="EULER")"""
    B = pose_src.shape[0]
    rot = _new((B, 3), pose_src)
    trans = _new((B, 3), pose_src)
    k1, mp = host_f32(T_means, 3)
    k2, sp = host_f32(T_stds, 3)
    check(lib().dim_se3_delta_euler(dptr(pose_src, f32), dptr(pose_tgt, f32), dptr(rot, f32), dptr(trans, f32), B,
                                    capi.rot_coord_id(rot_coord), mp, sp, current_stream()))
    return rot, trans


def transform3d_fwd(points, rot, trans, pose_src, rot_coord, T_means, T_stds, out=None):
    B = points.shape[0]
    npts = points.numel() // (B * 3) if B else 0
    out = out if out is not None else torch.empty_like(points)
    k1, mp = host_f32(T_means, 3)
    k2, sp = host_f32(T_stds, 3)
    check(lib().dim_transform3d_fwd(dptr(points, f32), dptr(rot, f32), dptr(trans, f32), dptr(pose_src, f32), dptr(out, f32), B, npts,
                                    capi.rot_coord_id(rot_coord), mp, sp, current_stream()))
    return out


def transform3d_bwd(out_grad, points, rot, trans, pose_src, rot_coord, T_means, T_stds):
    B = points.shape[0]
    npts = points.numel() // (B * 3) if B else 0
    d_rot = torch.empty_like(rot)
    d_trans = torch.empty_like(trans)
    k1, mp = host_f32(T_means, 3)
    k2, sp = host_f32(T_stds, 3)
    check(lib().dim_transform3d_bwd(dptr(out_grad, f32), dptr(points, f32), dptr(rot, f32), dptr(trans, f32), dptr(pose_src, f32),
                                    dptr(d_rot, f32), dptr(d_trans, f32), B, npts, capi.rot_coord_id(rot_coord), mp, sp,
                                    current_stream()))
    return d_rot, d_trans


def depth_to_flow(depth_src, depth_tgt, KT, Kinv, flow=None, valid=None):
    B, _, H, W = depth_src.shape
    flow = flow if flow is not None else _new((B, 2, H, W), depth_src)
    valid = valid if valid is not None else _new((B, 1, H, W), depth_src)
    keep, kp = host_f32(Kinv, 9)
    check(lib().dim_depth_to_flow(dptr(depth_src, f32), dptr(depth_tgt, f32), dptr(KT, f32), kp, B, H, W, dptr(flow, f32),
                                  dptr(valid, f32), current_stream()))
    return flow, valid


def flow_epe_sums(flow_pred, flow_gt, visible, depth_rendered, sums=None, accumulate=False, workspace=None):
    """per-sample sums of calc_EPE_one_pair (reference deepim/core/tester.py:719-736): sums (B,5) float64 =
    [epe_all, epe_viz, epe_vizbg, num_viz, num_vizbg]; flow_pred is rounded to float16 first as tester.py:485-487 stores it"""
    B, _, H, W = flow_pred.shape
    assert flow_gt.shape == flow_pred.shape and visible.shape == (B, 1, H, W) and depth_rendered.shape == (B, 1, H, W)
    if sums is None:
        assert not accumulate
        sums = torch.empty((B, 5), dtype=torch.float64, device=flow_pred.device)
    if workspace is None:
        workspace = torch.empty((lib().dim_flow_epe_workspace_bytes(B) // 8,), dtype=torch.float64, device=flow_pred.device)
    check(lib().dim_flow_epe_sums(dptr(flow_pred, f32), dptr(flow_gt, f32), dptr(visible, f32), dptr(depth_rendered, f32), B, H, W,
                                  dptr(workspace, torch.float64), dptr(sums, torch.float64), int(bool(accumulate)), current_stream()))
    return sums


def box_mask(bbox, mask, bbox_of_mask=None):
    """mask <- filled rectangle of bbox (end-exclusive); bbox_of_mask (B,4) int32: optional bbox of that rectangle"""
    B, _, H, W = mask.shape
    check(lib().dim_box_mask(dptr(bbox, i32), dptr(mask, f32), B, H, W, dptr(bbox_of_mask, i32) if bbox_of_mask is not None else None,
                             current_stream()))
    return mask


def test_blobs_from_raw(obs_bgr, ren_bgr, depth_ren, depth_factor, pixel_means_bgr, image_observed, image_rendered, mask_rendered, bbox,
                        mask_thr=0.2):
    """uint8 BGR images (B,H,W,3) + uint16 rendered depth (B,H,W) -> the float blobs of a test batch, on the device (csrc/data.hip)"""
    B, H, W = depth_ren.shape
    keep, mp = host_f32(pixel_means_bgr, 3)
    check(lib().dim_test_blobs_from_raw(dptr(obs_bgr, torch.uint8), dptr(ren_bgr, torch.uint8), dptr(depth_ren, torch.uint16), B, H, W,
                                        float(depth_factor), mp, float(mask_thr), dptr(image_observed, f32), dptr(image_rendered, f32),
                                        dptr(mask_rendered, f32), dptr(bbox, i32), current_stream()))


def pair_blobs_from_raw(B, H, W, depth_factor, pixel_means_bgr, obs_bgr=None, bg_bgr=None, use_bg=None, ren_bgr=None, depth_ren=None,
                        depth_a=None, depth_b=None, label=None, mask_idx=None, image_observed=None, image_rendered=None, mask_rendered=None,
                        depth_rendered=None, depth_a_out=None, depth_b_out=None, mask_label=None, label_raw=None, bbox_ren=None,
                        bbox_label=None, mask_thr=0.2):
    """raw file pixels of B pairs -> float blobs of a training / test batch on the device (csrc/data.hip; include/deepim_hip.h lists the
    rules); every tensor optional"""
    keep, mp = host_f32(pixel_means_bgr, 3)
    u8, u16 = torch.uint8, torch.uint16
    check(lib().dim_pair_blobs_from_raw(dptr(obs_bgr, u8), dptr(bg_bgr, u8), dptr(use_bg, i32), dptr(ren_bgr, u8), dptr(depth_ren, u16),
                                        dptr(depth_a, u16), dptr(depth_b, u16), dptr(label, u8), dptr(mask_idx, i32), B, H, W,
                                        float(depth_factor), mp, float(mask_thr), dptr(image_observed, f32), dptr(image_rendered, f32),
                                        dptr(mask_rendered, f32), dptr(depth_rendered, f32), dptr(depth_a_out, f32), dptr(depth_b_out, f32),
                                        dptr(mask_label, f32), dptr(label_raw, f32), dptr(bbox_ren, i32), dptr(bbox_label, i32),
                                        current_stream()))


def mask_dilate(mask_in, thickness4, out=None):
    """mask_dilate.py:10-55 with the caller's draws: thickness4 (B,4) int32 = {down, up, right, left}, 0 = side skipped"""
    B, _, H, W = mask_in.shape
    out = out if out is not None else torch.empty_like(mask_in)
    check(lib().dim_mask_dilate(dptr(mask_in, f32), dptr(thickness4, i32), dptr(out, f32), B, H, W, current_stream()))
    return out


FLOW_WEIGHT_ID = {"all": 0, "viz": 1, "valid": 2}


def calc_flow_labels(depth_src, depth_tgt, P12, Kinv_f64, flow, flow_weights=None, thresh=3e-3, standard_rep=False, weight_type="viz"):
    """first-iteration flow labels (calc_flow of lib/pair_matching/flow.py + the weights of get_pair_flow)"""
    B, _, H, W = depth_src.shape
    kinv = np.ascontiguousarray(np.asarray(Kinv_f64, dtype=np.float64).reshape(9))
    check(lib().dim_calc_flow_labels(dptr(depth_src, f32), dptr(depth_tgt, f32), dptr(P12, torch.float64), kinv.ctypes.data, B, H, W, float(thresh),
                                     int(bool(standard_rep)), FLOW_WEIGHT_ID[weight_type], dptr(flow, f32), dptr(flow_weights, f32),
                                     current_stream()))
    return flow, flow_weights


def point_clouds(table, table_off, idx, pose_observed, model, weights, observed):
    B, n = idx.shape
    check(lib().dim_point_clouds(dptr(table, f32), dptr(table_off, i32), dptr(idx, i32), dptr(pose_observed, f32), B, n, dptr(model, f32),
                                 dptr(weights, f32), dptr(observed, f32), current_stream()))


def conv2d_pack_weight(w_oihw, as_bf16=False):
    Cout, Cin, KH, KW = w_oihw.shape
    n = lib().dim_conv2d_packed_weight_floats(Cout, Cin, KH, KW)
    wp = torch.empty((n,), dtype=bf16 if as_bf16 else f32, device=w_oihw.device)
    if as_bf16:
        check(lib().dim_conv2d_pack_weight_bf16(dptr(w_oihw.contiguous(), f32), dptr(wp, bf16), Cout, Cout, Cin, KH, KW, current_stream()))
    else:
        check(lib().dim_conv2d_pack_weight(dptr(w_oihw.contiguous(), f32), dptr(wp, f32), Cout, Cin, KH, KW, current_stream()))
    return wp


def fc_pack_weight(w_out_in, C, H, W):
    Out = w_out_in.shape[0]
    wp = _new((Out * C * H * W,), w_out_in)
    check(lib().dim_fc_pack_weight(dptr(w_out_in.contiguous(), f32), dptr(wp, f32), Out, C, H, W, current_stream()))
    return wp


def conv_out_hw(H, W, KH, KW, stride, pad):
    return (H + 2 * pad - KH) // stride + 1, (W + 2 * pad - KW) // stride + 1


TILE_NAMES = {1: "128x128 (4 waves)", 2: "128x64", 3: "64x64", 4: "128x128 (8 waves)"}


def conv_auto_plan(M, Cout, nchunks, cin=32):
    """default (tile, splits) of a direct f32 layer when the caller does not autotune: dim_conv_auto_plan -- the ONE copy of the
    rule, shared with the C resident loop (csrc/refiner.hip)"""
    import ctypes

    tile, splits = ctypes.c_int(0), ctypes.c_int(0)
    check(lib().dim_conv_auto_plan(int(M), int(Cout), int(nchunks), int(cin), ctypes.byref(tile), ctypes.byref(splits)))
    return tile.value, splits.value


def copy(dst, src):
    """dst <- src (same shape, 4-byte dtype, contiguous CUDA tensors) as a kernel launch: graph-safe replacement for Tensor.copy_
    (hipMemcpyAsync nodes inside captured graphs are avoided, see include/deepim_hip.h dim_copy_words)."""
    assert dst.shape == src.shape and dst.dtype == src.dtype and dst.element_size() == 4, (dst.shape, src.shape, dst.dtype, src.dtype)
    check(lib().dim_copy_words(dptr(dst), dptr(src), dst.numel(), current_stream()))
    return dst


def copy_nhwc_channels(dst, dst_c0, src, src_c0, nch, add=False):
    """dst[..., dst_c0:dst_c0+nch] (+)= src[..., src_c0:src_c0+nch] for two contiguous f32 tensors with the same leading dims and their own
    channel counts (NHWC maps, or (B, C) matrices): dim_copy_rows / dim_add_rows with rows = pixels -- the strided copy / add a
    Tensor.copy_ / add_ on a channel slice would hand to a vendor elementwise kernel"""
    assert dst.is_contiguous() and src.is_contiguous() and dst.shape[:-1] == src.shape[:-1], (dst.shape, src.shape)
    Cd, Cs = dst.shape[-1], src.shape[-1]
    assert 0 <= dst_c0 and dst_c0 + nch <= Cd and 0 <= src_c0 and src_c0 + nch <= Cs
    rows = dst.numel() // Cd
    fn = lib().dim_add_rows if add else lib().dim_copy_rows
    check(fn(dptr(dst, f32) + 4 * dst_c0, Cd, dptr(src, f32) + 4 * src_c0, Cs, rows, nch, current_stream()))
    return dst


def fill(dst, value=0.0):
    """dst[...] = value (contiguous f32 / int32 tensor) as a kernel launch (dim_fill_words)"""
    assert dst.is_contiguous() and dst.element_size() == 4
    bits = struct.unpack("<I", struct.pack("<f", float(value)))[0] if dst.dtype == f32 else int(value) & 0xFFFFFFFF
    check(lib().dim_fill_words(dptr(dst), dst.numel(), bits, current_stream()))
    return dst


def copy_channels(dst_oihw, dst_c0, src_oihw, src_c0, nch):
    """dst[:, dst_c0:dst_c0+nch] <- src[:, src_c0:src_c0+nch] for two contiguous (O, I, kh, kw) f32 weights with the same O, kh, kw
    (dim_copy_rows: one run of nch*kh*kw floats per output channel)"""
    O, Id, kh, kw = dst_oihw.shape
    Os, Is = src_oihw.shape[:2]
    assert O == Os and tuple(src_oihw.shape[2:]) == (kh, kw) and dst_oihw.is_contiguous() and src_oihw.is_contiguous()
    assert 0 <= dst_c0 and dst_c0 + nch <= Id and 0 <= src_c0 and src_c0 + nch <= Is
    t = kh * kw
    check(lib().dim_copy_rows(dptr(dst_oihw, f32) + 4 * dst_c0 * t, Id * t, dptr(src_oihw, f32) + 4 * src_c0 * t, Is * t, O, nch * t,
                              current_stream()))
    return dst_oihw


def set_winograd_split(on):
    """arithmetic of the Winograd layers' plane GEMMs: True = f32 operands as three bf16 terms, six MFMA products (default);
    False = the f32 matrix pipe.  Read when a layer is next planned (launch / graph capture)."""
    check(lib().dim_set_winograd_split(1 if on else 0))


def get_winograd_split():
    return bool(lib().dim_get_winograd_split())


def winograd_pack_weight(w_oihw, m=2):
    """(Cout,Cin,3,3) -> the (m+2)^2 transformed 1x1 weight sets of the Winograd F(m x m, 3x3) path, m = 2 or 4"""
    Cout, Cin, KH, KW = w_oihw.shape
    assert KH == 3 and KW == 3
    out = _new((lib().dim_winograd_packed_weight_floats(Cout, Cin, m),), w_oihw)
    check(lib().dim_winograd_pack_weight(dptr(w_oihw, f32), dptr(out, f32), Cout, Cin, m, current_stream()))
    return out


def winograd_dgrad_pack_weight(w_oihw, m=2):
    """the transformed weights of the layer's INPUT gradient from its forward (Cout,Cin,3,3) array: the same array
    winograd_pack_weight(w.flip(2, 3).transpose(0, 1)) gives, without the flipped copy"""
    Cout, Cin, KH, KW = w_oihw.shape
    assert KH == 3 and KW == 3 and w_oihw.is_contiguous()
    out = _new((lib().dim_winograd_packed_weight_floats(Cin, Cout, m),), w_oihw)
    check(lib().dim_winograd_dgrad_pack_weight(dptr(w_oihw, f32), dptr(out, f32), Cout, Cin, m, current_stream()))
    return out


def conv2d_fwd_winograd(x_nhwc, Cin, w_packed, bias, Cout, slope=0.1, tile=0, out=None, out_coff=0, workspace=None, events=None, m=2):
    """3x3 / stride 1 / pad 1 convolution through Winograd F(m x m, 3x3) (m = 2 or 4, the m the weights were packed with);
    x may carry padded channels (in_cstride = x.shape[-1]).
    events: optional list; ("wino_in" | "conv" | "wino_out", start, end) HIP-event triples of the three launches are appended."""
    import ctypes

    ev_arr, evs = None, None
    if events is not None:
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
        for e in evs:
            e.record()  # materialises the hipEvent_t behind the torch object
        ev_arr = (ctypes.c_void_p * 4)(*[e.cuda_event for e in evs])
    N, H, W, in_cs = x_nhwc.shape
    out = out if out is not None else _new((N, H, W, Cout), x_nhwc)
    need = lib().dim_winograd_workspace_floats(N, H, W, Cin, Cout, m)
    if workspace is None or workspace.numel() < need:
        workspace = _new((need,), x_nhwc)
    check(lib().dim_conv2d_fwd_winograd(dptr(x_nhwc, f32), dptr(w_packed, f32), dptr(bias, f32), dptr(out, f32), dptr(workspace, f32), N, H, W,
                                        Cin, in_cs, Cout, out.shape[-1], out_coff, float(slope), tile, m, ev_arr, current_stream()))
    if events is not None:
        events.extend([("wino_in", evs[0], evs[1]), ("conv", evs[1], evs[2]), ("wino_out", evs[2], evs[3])])
    return out



def winograd3x3s2_pack_weight(w_oihw):
    """(Cout,Cin,3,3) of a stride-2 / pad-1 layer -> the 81 transformed weight sets of dim_conv2d_fwd_winograd3x3s2"""
    Cout, Cin, KH, KW = w_oihw.shape
    assert KH == 3 and KW == 3
    out = _new((lib().dim_winograd3x3s2_packed_weight_floats(Cout, Cin),), w_oihw)
    check(lib().dim_winograd3x3s2_pack_weight(dptr(w_oihw.contiguous(), f32), dptr(out, f32), Cout, Cin, current_stream()))
    return out


def conv2d_fwd_winograd3x3s2(x_nhwc, Cin, w_packed, bias, Cout, slope=0.1, tile=0, out=None, out_coff=0, workspace=None, events=None):
    """y = LeakyReLU(conv 3x3 / stride 2 / pad 1 + bias) through the phase-image Winograd path (81 plane GEMMs); x (N,H,W,>=Cin) NHWC.
    events: as conv2d_fwd_winograd."""
    N, H, W, in_cs = x_nhwc.shape
    out = out if out is not None else _new((N, (H - 1) // 2 + 1, (W - 1) // 2 + 1, Cout), x_nhwc)
    need = lib().dim_winograd3x3s2_workspace_floats(N, H, W, Cin, Cout)
    if workspace is None or workspace.numel() < need:
        workspace = _new((need,), x_nhwc)
    ev_arr, evs = _wino_events(events)
    check(lib().dim_conv2d_fwd_winograd3x3s2(dptr(x_nhwc, f32), dptr(w_packed, f32), dptr(bias, f32), dptr(out, f32), dptr(workspace, f32),
                                             N, H, W, Cin, in_cs, Cout, out.shape[-1], out_coff, float(slope), tile, ev_arr, current_stream()))
    if events is not None:
        events.extend([("wino_in", evs[0], evs[1]), ("conv", evs[1], evs[2]), ("wino_out", evs[2], evs[3])])
    return out


def winograd5x5s2_pack_weight(w_oihw):
    """(Cout,Cin,5,5) of a stride-2 / pad-2 layer -> the 36 transformed weight sets (K = 4 Cin) of dim_conv2d_fwd_winograd5x5s2"""
    Cout, Cin, KH, KW = w_oihw.shape
    assert KH == 5 and KW == 5
    out = _new((lib().dim_winograd5x5s2_packed_weight_floats(Cout, Cin),), w_oihw)
    check(lib().dim_winograd5x5s2_pack_weight(dptr(w_oihw, f32), dptr(out, f32), Cout, Cin, current_stream()))
    return out


def winograd5x5s2_dgrad_pack_weight(w_oihw):
    """(Cout,Cin,5,5) -> the 36 transformed weight sets (K = Cout, N = 4 Cin) of dim_conv2d_dgrad_winograd5x5s2"""
    Cout, Cin, KH, KW = w_oihw.shape
    assert KH == 5 and KW == 5
    out = _new((lib().dim_winograd5x5s2_packed_weight_floats(Cout, Cin),), w_oihw)
    check(lib().dim_winograd5x5s2_dgrad_pack_weight(dptr(w_oihw, f32), dptr(out, f32), Cout, Cin, current_stream()))
    return out


def conv2d_dgrad_winograd5x5s2(dy_nhwc, Cout, w_packed, dx_nhwc, Cin, tile=0, workspace=None):
    """dx (N,H,W,>=Cin) = input gradient of the 5x5 / stride 2 / pad 2 convolution, from dy (N,ceil(H/2),ceil(W/2),>=Cout); overwrites dx"""
    N, H, W, dx_cs = dx_nhwc.shape
    assert dy_nhwc.shape[:3] == (N, (H + 1) // 2, (W + 1) // 2)
    need = lib().dim_winograd5x5s2_workspace_floats(N, H, W, Cin, Cout)
    if workspace is None or workspace.numel() < need:
        workspace = _new((need,), dx_nhwc)
    check(lib().dim_conv2d_dgrad_winograd5x5s2(dptr(dy_nhwc, f32), dptr(w_packed, f32), dptr(dx_nhwc, f32), dptr(workspace, f32), N, H, W, Cin,
                                               dx_cs, Cout, dy_nhwc.shape[-1], tile, current_stream()))
    return dx_nhwc


def conv2d_wgrad_winograd(x_nhwc, Cin, dy_nhwc, Cout, dw_oihw, S=1, splits=4, workspace=None, scale=1.0, accumulate=False):
    """dw (Cout,Cin,k,k) (+)= scale * weight gradient through Winograd; S = 1: 3x3 / stride 1 / pad 1, S = 2: 5x5 / stride 2 / pad 2"""
    N, H, W, in_cs = x_nhwc.shape
    k = 3 if S == 1 else 5
    assert tuple(dw_oihw.shape) == (Cout, Cin, k, k) and dw_oihw.is_contiguous()
    need = lib().dim_conv2d_wgrad_winograd_workspace_floats(N, H, W, Cin, Cout, S, splits)
    if workspace is None or workspace.numel() < need:
        workspace = _new((need,), x_nhwc)
    check(lib().dim_conv2d_wgrad_winograd(dptr(x_nhwc, f32), dptr(dy_nhwc, f32), dptr(dw_oihw, f32), dptr(workspace, f32), N, H, W, Cin, in_cs,
                                          Cout, dy_nhwc.shape[-1], S, splits, float(scale), int(accumulate), current_stream()))
    return dw_oihw


def fc_fwd(x_nhwc, w_packed, bias, Out, slope=0.1, out=None, workspace=None, events=None):
    """y (B,Out) = LeakyReLU(flatten(x) . W^T + b) as a weight stream (dim_fc_fwd); w_packed from fc_pack_weight.
    events: optional list; one ("fc", start, end, 2) HIP-event tuple around the stream + reduce launches is appended."""
    B, H, W, C = x_nhwc.shape
    assert x_nhwc.is_contiguous()
    out = out if out is not None else _new((B, Out), x_nhwc)
    need = lib().dim_fc_fwd_workspace_floats(C, H, W, Out)
    if workspace is None or workspace.numel() < need:
        workspace = _new((need,), x_nhwc)
    evs = None
    if events is not None:
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        evs[0].record()
    check(lib().dim_fc_fwd(dptr(x_nhwc, f32), dptr(w_packed, f32), dptr(bias, f32), dptr(out, f32), dptr(workspace, f32), B, C, H, W, Out,
                           float(slope), current_stream()))
    if events is not None:
        evs[1].record()
        events.append(("fc", evs[0], evs[1], 2))
    return out


def _wino_events(events):
    import ctypes

    if events is None:
        return None, None
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    for e in evs:
        e.record()  # materialises the hipEvent_t behind the torch object
    return (ctypes.c_void_p * 4)(*[e.cuda_event for e in evs]), evs


def conv2d_fwd_winograd5x5s2(x_nhwc, Cin, w_packed, bias, Cout, slope=0.1, tile=0, out=None, out_coff=0, workspace=None, events=None):
    """5x5 / stride 2 / pad 2 convolution as four phase images through Winograd F(4x4,3x3) (see include/deepim_hip.h)."""
    ev_arr, evs = _wino_events(events)
    N, H, W, in_cs = x_nhwc.shape
    out = out if out is not None else _new((N, (H + 1) // 2, (W + 1) // 2, Cout), x_nhwc)
    need = lib().dim_winograd5x5s2_workspace_floats(N, H, W, Cin, Cout)
    if workspace is None or workspace.numel() < need:
        workspace = _new((need,), x_nhwc)
    check(lib().dim_conv2d_fwd_winograd5x5s2(dptr(x_nhwc, f32), dptr(w_packed, f32), dptr(bias, f32), dptr(out, f32), dptr(workspace, f32),
                                             N, H, W, Cin, in_cs, Cout, out.shape[-1], out_coff, float(slope), tile, ev_arr, current_stream()))
    if events is not None:
        events.extend([("wino_in", evs[0], evs[1]), ("conv", evs[1], evs[2]), ("wino_out", evs[2], evs[3])])
    return out

def conv2d_fwd(x_nhwc, w_packed, bias, Cout, KH, KW, stride, pad, slope=0.1, splits=1, tile=0, out=None, workspace=None,
               events=None):
    """splits: 1 = one launch; > 1 = split-K through `workspace`; 0 = auto (whole tiles per CU in one launch, the remaining tiles
    as a split-K launch + reduce; needs `workspace`, see dim_conv2d_fwd / dim_conv2d_tail_plan).
    events: optional list; when given, (kernel_tag, start_event, end_event) tuples are appended with HIP events recorded on the
    launch stream around the conv kernel and (split-K) around the reduce kernel separately (auto mode: one tuple for everything,
    tag "conv", with a 4th element = number of conv-kernel launches inside)."""
    N, H, W, Cin = x_nhwc.shape
    Ho, Wo = conv_out_hw(H, W, KH, KW, stride, pad)
    out = out if out is not None else _new((N, Ho, Wo, Cout), x_nhwc)
    if splits != 1 and workspace is None:
        workspace = _new((lib().dim_conv2d_workspace_floats(N, H, W, Cin, Cout, KH, KW, stride, pad, splits),), x_nhwc)
    if events is not None:
        def ev():
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            return e
        e0 = ev()
        if splits > 1 and w_packed.dtype != bf16:
            check(lib().dim_conv2d_fwd_partial(dptr(x_nhwc, f32), dptr(w_packed, f32), dptr(workspace, f32), N, H, W, Cin, Cout, KH, KW,
                                               stride, pad, splits, tile, current_stream()))
            e1 = ev()
            check(lib().dim_splitk_reduce(dptr(workspace, f32), dptr(bias, f32), dptr(out, f32), N * Ho * Wo, Cout, splits, float(slope),
                                          current_stream()))
            e2 = ev()
            events += [("conv", e0, e1), ("reduce", e1, e2)]
        else:
            wp, is16 = _wp(w_packed)
            check((lib().dim_conv2d_fwd_bf16 if is16 else lib().dim_conv2d_fwd)(dptr(x_nhwc, f32), wp, dptr(bias, f32), dptr(out, f32),
                                                                                 dptr(workspace, f32), N, H, W, Cin, Cout, KH, KW, stride, pad,
                                                                                 float(slope), splits, tile, current_stream()))
            n_launch = 1
            if splits == 0:
                import ctypes

                tb, ts = ctypes.c_int(0), ctypes.c_int(1)
                check(lib().dim_conv2d_tail_plan(N * Ho * Wo, Cout, Cin, KH, KW, tile, ctypes.byref(tb), ctypes.byref(ts)))
                n_launch = 2 if ts.value >= 2 else 1
            events.append(("conv", e0, ev(), n_launch))
        return out
    wp, is16 = _wp(w_packed)
    fn = lib().dim_conv2d_fwd_bf16 if is16 else lib().dim_conv2d_fwd
    check(fn(dptr(x_nhwc, f32), wp, dptr(bias, f32), dptr(out, f32), dptr(workspace, f32), N, H, W,
             Cin, Cout, KH, KW, stride, pad, float(slope), splits, tile, current_stream()))
    return out


def pose_head_fwd(fc6, p, zf, se3=None, fc7_out=None):
    """p: dict with fc7_weight/bias, rot_weight/bias, trans_weight/bias (reference (out,in) layout)."""
    B = fc6.shape[0]
    se3 = se3 if se3 is not None else _new((B, 7), fc6)
    check(lib().dim_pose_head_fwd(dptr(fc6, f32), dptr(p["fc7_weight"], f32), dptr(p["fc7_bias"], f32), dptr(p["rot_weight"], f32),
                                  dptr(p["rot_bias"], f32), dptr(p["trans_weight"], f32), dptr(p["trans_bias"], f32), dptr(zf, f32),
                                  dptr(se3, f32), dptr(fc7_out, f32), B, current_stream()))
    return se3


# ---------------------------------------------------------------- decoder pieces (deepIM_flownet.py:213-299, :315-340, :502-529)
def pad32(c):
    return (c + 31) // 32 * 32


def _new_packed(n, like, as_bf16):
    return torch.empty((n,), dtype=bf16 if as_bf16 else f32, device=like.device)


def deconv4x4s2_pack_weight(w_iohw, as_bf16=False):
    """as_bf16: the bf16 image of the packed array in one pass (== to_bf16 of the f32 result); same for the packers below"""
    Cin, Cout = w_iohw.shape[:2]
    wp = _new_packed(lib().dim_deconv4x4s2_packed_weight_floats(Cin, Cout), w_iohw, as_bf16)
    if as_bf16:
        check(lib().dim_deconv4x4s2_pack_weight_bf16(dptr(w_iohw.contiguous(), f32), dptr(wp, bf16), Cin, Cout, current_stream()))
    else:
        check(lib().dim_deconv4x4s2_pack_weight(dptr(w_iohw.contiguous(), f32), dptr(wp, f32), Cin, Cout, current_stream()))
    return wp


def deconv4x4s2_fwd(x_nhwc, Cin, w_packed, bias, y_nhwc, Cout, crop, slope, out_coff=0, tile=3):
    """y[..., out_coff:out_coff+Cout] = LeakyReLU(Crop(Deconvolution(x[..., :Cin]) + bias)); y's H,W define the crop window."""
    N, H, W, in_cs = x_nhwc.shape
    _, OH, OW, out_cs = y_nhwc.shape
    wp, is16 = _wp(w_packed)
    fn = lib().dim_deconv4x4s2_fwd_bf16 if is16 else lib().dim_deconv4x4s2_fwd
    check(fn(dptr(x_nhwc, f32), wp, dptr(bias, f32), dptr(y_nhwc, f32), N, H, W, Cin, in_cs, Cout,
             OH, OW, crop, float(slope), out_cs, out_coff, tile, current_stream()))
    return y_nhwc


def deconv4x4s2_tiny_fwd(x_nhwc, Cin, w_iohw, bias, y_nhwc, Cout, crop, out_coff=0):
    N, H, W, in_cs = x_nhwc.shape
    _, OH, OW, out_cs = y_nhwc.shape
    check(lib().dim_deconv4x4s2_tiny_fwd(dptr(x_nhwc, f32), dptr(w_iohw, f32), dptr(bias, f32), dptr(y_nhwc, f32), N, H, W, Cin, in_cs,
                                         Cout, OH, OW, crop, out_cs, out_coff, current_stream()))
    return y_nhwc


def conv_small_cout_pack_weight(w_oihw):
    Cout, Cin, KH, KW = w_oihw.shape
    wp = _new((Cout * KH * KW * pad32(Cin),), w_oihw)
    check(lib().dim_conv_small_cout_pack_weight(dptr(w_oihw.contiguous(), f32), dptr(wp, f32), Cout, Cin, KH, KW, current_stream()))
    return wp


def conv_small_cout_fwd(x_nhwc, Cin, w_packed, bias, Cout, KH=3, KW=3, pad=1, out=None, out_coff=0):
    N, H, W, in_cs = x_nhwc.shape
    out = out if out is not None else _new((N, H, W, Cout), x_nhwc)
    check(lib().dim_conv_small_cout_fwd(dptr(x_nhwc, f32), dptr(w_packed, f32), dptr(bias, f32), dptr(out, f32), N, H, W, Cin, in_cs, Cout,
                                        KH, KW, pad, out.shape[3], out_coff, current_stream()))
    return out


def upsample16_fwd(x_nhwc, w_c1_32_32, OH, OW, crop=8, scale=1.0, sigmoid=False, out=None):
    N, h, w, C = x_nhwc.shape
    out = out if out is not None else _new((N, C, OH, OW), x_nhwc)
    check(lib().dim_upsample16_fwd(dptr(x_nhwc, f32), dptr(w_c1_32_32, f32), dptr(out, f32), N, C, h, w, OH, OW, crop, float(scale),
                                   1 if sigmoid else 0, current_stream()))
    return out


# ---------------------------------------------------------------- backward of the convolution stack
def pad64(c):
    return (c + 63) // 64 * 64


def conv2d_dgrad_pack_weight(w_oihw, stride, pad, as_bf16=False):
    Cout, Cin, KH, KW = w_oihw.shape
    wp = _new_packed(lib().dim_conv2d_dgrad_packed_weight_floats(Cout, Cin, KH, KW, stride, pad), w_oihw, as_bf16)
    if as_bf16:
        check(lib().dim_conv2d_dgrad_pack_weight_bf16(dptr(w_oihw.contiguous(), f32), dptr(wp, bf16), Cout, Cin, KH, KW, stride, pad,
                                                      current_stream()))
    else:
        check(lib().dim_conv2d_dgrad_pack_weight(dptr(w_oihw.contiguous(), f32), dptr(wp, f32), Cout, Cin, KH, KW, stride, pad, current_stream()))
    return wp


def conv2d_dgrad_lrelu(dy_nhwc, Cout, w_dgrad_packed_bf16, dz_nhwc, y_act_nhwc, Cin, KH, KW, stride, pad, db, slope=0.1, workspace=None,
                       accumulate_db=False):
    """dz = dgrad(dy) * LeakyReLU'(y_act), db (+)= column sums of dz -- the lower layer's activation gradient pass folded into this
    layer's input gradient (dim_conv2d_dgrad_bf16_lrelu: bf16 patch kernel only).  dz, y_act: dense (N, H, W, Cin)."""
    N, Ho, Wo, dy_cs = dy_nhwc.shape
    _, H, W, dx_cs = dz_nhwc.shape
    assert w_dgrad_packed_bf16.dtype == bf16 and tuple(y_act_nhwc.shape) == tuple(dz_nhwc.shape) == (N, H, W, Cin)
    assert dz_nhwc.is_contiguous() and y_act_nhwc.is_contiguous() and db.numel() == Cin
    need = lib().dim_conv2d_dgrad_lrelu_workspace_floats(N, H, W, Cin, stride)
    if workspace is None or workspace.numel() < need:
        workspace = _new((need,), dz_nhwc)
    check(lib().dim_conv2d_dgrad_bf16_lrelu(dptr(dy_nhwc, f32), dptr(w_dgrad_packed_bf16, bf16), dptr(dz_nhwc, f32), dptr(y_act_nhwc, f32),
                                            float(slope), dptr(db, f32), dptr(workspace, f32), N, H, W, Cin, dx_cs, Ho, Wo, Cout, dy_cs, KH, KW,
                                            stride, pad, int(accumulate_db), current_stream()))
    return dz_nhwc


def conv2d_dgrad(dy_nhwc, Cout, w_dgrad_packed, dx_nhwc, Cin, KH, KW, stride, pad, accumulate=False, tile=3, splits=1, workspace=None):
    """dx[..., :Cin] (+)= dgrad(dy[..., :Cout]); dx / dy may be wider concat buffers.
    splits > 1 (bf16 weights, tile 3 / 4): split-K through output-shaped slabs in `workspace` (small maps: too few tiles for the chip)"""
    N, Ho, Wo, dy_cs = dy_nhwc.shape
    _, H, W, dx_cs = dx_nhwc.shape
    wp, is16 = _wp(w_dgrad_packed)
    if splits > 1:
        assert is16, "split-K input gradient is built for the bf16 kernels"
        need = lib().dim_conv2d_dgrad_splitk_workspace_floats(N, H, W, dx_cs, splits)
        if workspace is None or workspace.numel() < need:
            workspace = _new((need,), dx_nhwc)
        check(lib().dim_conv2d_dgrad_bf16_splitk(dptr(dy_nhwc, f32), wp, dptr(dx_nhwc, f32), dptr(workspace, f32), N, H, W, Cin, dx_cs, Ho, Wo,
                                                 Cout, dy_cs, KH, KW, stride, pad, int(accumulate), tile, splits, current_stream()))
        return dx_nhwc
    fn = lib().dim_conv2d_dgrad_bf16 if is16 else lib().dim_conv2d_dgrad
    check(fn(dptr(dy_nhwc, f32), wp, dptr(dx_nhwc, f32), N, H, W, Cin, dx_cs, Ho, Wo, Cout, dy_cs,
             KH, KW, stride, pad, int(accumulate), tile, current_stream()))
    return dx_nhwc


def conv2d_wgrad(x_nhwc, Cin, dz_nhwc, Cout, KH, KW, stride, pad, dw_packed, splits=1, dz_coff=0, workspace=None, accumulate=False,
                 bf16_mfma=False):
    """bf16_mfma: products on the bf16 matrix pipe (operands rounded on the way into LDS; f32 accumulate, f32 result)"""
    N, H, W, in_cs = x_nhwc.shape
    _, Ho, Wo, dz_cs = dz_nhwc.shape
    if splits > 1 and workspace is None:
        workspace = _new((lib().dim_conv2d_wgrad_workspace_floats(Cout, Cin, KH, KW, splits),), x_nhwc)
    check((lib().dim_conv2d_wgrad_bf16 if bf16_mfma else lib().dim_conv2d_wgrad)(dptr(x_nhwc, f32), dptr(dz_nhwc, f32), dptr(dw_packed, f32), dptr(workspace, f32), N, H, W, Cin, in_cs, Ho,
                                 Wo, Cout, dz_cs, dz_coff, KH, KW, stride, pad, splits, int(accumulate), current_stream()))
    return dw_packed


def conv2d_wgrad_oihw(x_nhwc, Cin, dz_nhwc, Cout, KH, KW, stride, pad, dw_oihw, splits=1, dz_coff=0, workspace=None, bf16_mfma=False,
                      scale=1.0, accumulate=False):
    """dw_oihw (rows <= Cout, Cin, KH, KW) (+)= scale * weight gradient, slabs summed inside the layout converter (dim_conv2d_wgrad_oihw)"""
    N, H, W, in_cs = x_nhwc.shape
    _, Ho, Wo, dz_cs = dz_nhwc.shape
    assert dw_oihw.is_contiguous() and tuple(dw_oihw.shape[1:]) == (Cin, KH, KW) and dw_oihw.shape[0] <= Cout
    need = lib().dim_conv2d_wgrad_workspace_floats(Cout, Cin, KH, KW, max(int(splits), 1) + 1)
    if workspace is None or workspace.numel() < need:
        workspace = _new((need,), x_nhwc)
    check(lib().dim_conv2d_wgrad_oihw(dptr(x_nhwc, f32), dptr(dz_nhwc, f32), dptr(dw_oihw, f32), dptr(workspace, f32), N, H, W, Cin, in_cs, Ho, Wo,
                                      Cout, dz_cs, dz_coff, KH, KW, stride, pad, int(splits), int(bool(bf16_mfma)), dw_oihw.shape[0],
                                      float(scale), int(accumulate), current_stream()))
    return dw_oihw


def bias_grad(dz_nhwc, C, db, dz_coff=0, workspace=None, accumulate=False):
    M = dz_nhwc.numel() // dz_nhwc.shape[-1]
    if workspace is None:
        workspace = _new((lib().dim_bias_grad_workspace_floats(M, C),), dz_nhwc)
    check(lib().dim_bias_grad(dptr(dz_nhwc, f32), dptr(db, f32), dptr(workspace, f32), M, C, dz_nhwc.shape[-1], dz_coff, int(accumulate),
                              current_stream()))
    return db


def lrelu_bwd(y_nhwc, dy_nhwc, C, slope=0.1, y_coff=0, dy_coff=0):
    M = dy_nhwc.numel() // dy_nhwc.shape[-1]
    check(lib().dim_lrelu_bwd(dptr(y_nhwc, f32), y_nhwc.shape[-1], y_coff, dptr(dy_nhwc, f32), dy_nhwc.shape[-1], dy_coff, M, C, float(slope),
                              current_stream()))
    return dy_nhwc


def lrelu_bwd_bias_grad(y_nhwc, dy_nhwc, C, db, slope=0.1, y_coff=0, dy_coff=0, workspace=None, accumulate=False):
    """dy[..., dy_coff:dy_coff+C] *= LeakyReLU'(y[..., y_coff:y_coff+C]) in place and db (+)= its column sums, one pass (dim_lrelu_bwd_bias_grad)"""
    M = dy_nhwc.numel() // dy_nhwc.shape[-1]
    need = lib().dim_lrelu_bwd_bias_grad_workspace_floats(M, C)
    if workspace is None or workspace.numel() < need:
        workspace = _new((need,), dy_nhwc)
    check(lib().dim_lrelu_bwd_bias_grad(dptr(y_nhwc, f32), y_nhwc.shape[-1], y_coff, dptr(dy_nhwc, f32), dy_nhwc.shape[-1], dy_coff, dptr(db, f32),
                                        dptr(workspace, f32), M, C, float(slope), int(accumulate), current_stream()))
    return dy_nhwc


# ---------------------------------------------------------------- training-only pieces (csrc/train.hip)
def _p(t, off=0):
    """device pointer of a contiguous CUDA f32 tensor, advanced by `off` floats (channel offset inside an NHWC row)"""
    return dptr(t, f32) + 4 * off


def conv2d_pack_weight_padded(w_oihw, CoutPad, as_bf16=False):
    Cout, Cin, KH, KW = w_oihw.shape
    wp = _new_packed(KH * KW * Cin * CoutPad, w_oihw, as_bf16)
    if as_bf16:
        check(lib().dim_conv2d_pack_weight_bf16(dptr(w_oihw.contiguous(), f32), dptr(wp, bf16), Cout, CoutPad, Cin, KH, KW, current_stream()))
    else:
        check(lib().dim_conv2d_pack_weight_padded(dptr(w_oihw.contiguous(), f32), dptr(wp, f32), Cout, CoutPad, Cin, KH, KW, current_stream()))
    return wp


def conv2d_unpack_weight(w_packed, out_oihw, CoutPad=None, scale=1.0, accumulate=False):
    Cout, Cin, KH, KW = out_oihw.shape
    check(lib().dim_conv2d_unpack_weight(dptr(w_packed, f32), dptr(out_oihw, f32), Cout, CoutPad or Cout, Cin, KH, KW, float(scale),
                                         int(accumulate), current_stream()))
    return out_oihw


def fc_unpack_weight(w_packed, out_w, C, H, W):
    check(lib().dim_fc_unpack_weight(dptr(w_packed, f32), dptr(out_w, f32), out_w.shape[0], C, H, W, current_stream()))
    return out_w


def fc_wgrad_nhwc(dz, x_nhwc, dW):
    """dW (Out, C*H*W in MXNet's (c, h, w) order) = dz (B, Out)^T . x (B, H, W, C); B <= 32 (dim_fc_wgrad_nhwc)"""
    B, H, W, C = x_nhwc.shape
    Out = dW.shape[0]
    assert x_nhwc.is_contiguous() and dW.is_contiguous() and dW.numel() == Out * C * H * W and dz.numel() == B * Out
    check(lib().dim_fc_wgrad_nhwc(dptr(dz, f32), dptr(x_nhwc, f32), dptr(dW, f32), B, Out, C, H, W, current_stream()))
    return dW


def fc_dgrad_pack_weight(w_out_in, C, H, W, out=None, as_bf16=False):
    Out = w_out_in.shape[0]
    out = out if out is not None else _new_packed(Out * C * H * W, w_out_in, as_bf16)
    if as_bf16:
        check(lib().dim_fc_dgrad_pack_weight_bf16(dptr(w_out_in, f32), dptr(out, bf16), Out, C, H, W, current_stream()))
    else:
        check(lib().dim_fc_dgrad_pack_weight(dptr(w_out_in, f32), dptr(out, f32), Out, C, H, W, current_stream()))
    return out


def conv2d_fwd_ex(x, x_coff, Cin, w_packed, bias, y, y_coff, Cout, KH, KW, stride, pad, slope=1.0, tile=3, Ho=0, Wo=0, accumulate=False):
    """dense generalised conv: channel windows [x_coff, x_coff+Cin) of x and [y_coff, y_coff+Cout) of y (NHWC concat buffers)."""
    N, H, W, in_cs = x.shape
    out_cs = y.shape[-1]
    wp, is16 = _wp(w_packed)
    fn = lib().dim_conv2d_fwd_ex_bf16 if is16 else lib().dim_conv2d_fwd_ex
    check(fn(_p(x, x_coff), wp, dptr(bias, f32), dptr(y, f32), N, H, W, Cin, in_cs, Cout, KH, KW, stride,
             pad, float(slope), tile, out_cs, y_coff, 0, 0, 0, 0, 0, 0, Ho, Wo, -1, int(accumulate), current_stream()))
    return y


def conv2d_wgrad_ex(x, x_coff, Cin, dz, dz_coff, Cout, KH, KW, stride, pad, dw_packed, splits=1, workspace=None, bf16_mfma=False):
    N, H, W, in_cs = x.shape
    _, Ho, Wo, dz_cs = dz.shape
    if splits > 1 and workspace is None:
        workspace = _new((lib().dim_conv2d_wgrad_workspace_floats(Cout, Cin, KH, KW, splits),), x)
    check((lib().dim_conv2d_wgrad_bf16 if bf16_mfma else lib().dim_conv2d_wgrad)(_p(x, x_coff), dptr(dz, f32), dptr(dw_packed, f32), dptr(workspace, f32), N, H, W, Cin, in_cs, Ho, Wo, Cout,
                                 dz_cs, dz_coff, KH, KW, stride, pad, splits, 0, current_stream()))
    return dw_packed


def flow_loss_grad(flow_est, flow_label, flow_weights, grad, normalize_flow, grad_scale, loss_sum=None):
    check(lib().dim_flow_loss_grad(dptr(flow_est, f32), dptr(flow_label, f32), dptr(flow_weights, f32), dptr(grad, f32), flow_est.numel(),
                                   float(normalize_flow), float(grad_scale), dptr(loss_sum, f32), current_stream()))
    return grad


def logistic_grad(logits, label, grad, grad_scale_over_num_output, prob=None):
    check(lib().dim_logistic_grad(dptr(logits, f32), dptr(label, f32), dptr(grad, f32), dptr(prob, f32), logits.numel(),
                                  float(grad_scale_over_num_output), current_stream()))
    return grad


def pm_l1_grad(p_est, p_obs, weights, grad, norm_term, grad_scale, loss_sum=None):
    check(lib().dim_pm_l1_grad(dptr(p_est, f32), dptr(p_obs, f32), dptr(weights, f32), dptr(grad, f32), p_est.numel(), float(norm_term),
                               float(grad_scale), dptr(loss_sum, f32), current_stream()))
    return grad


LOSS_TYPE_ID = {"L1": 0, "L2": 1, "smooth_L1": 2}


def pm_loss_grad(p_est, p_obs, weights, grad, norm_term, grad_scale, loss_type="L1", smooth_l1_scalar=1.0, loss_sum=None):
    """point-matching loss gradient for SE3_PM_LOSS_TYPE 'L1' | 'L2' | 'smooth_L1' (deepIM_flownet.py:458-499)"""
    check(lib().dim_pm_loss_grad(dptr(p_est, f32), dptr(p_obs, f32), dptr(weights, f32), dptr(grad, f32), p_est.numel(), float(norm_term),
                                 float(grad_scale), LOSS_TYPE_ID[loss_type], float(smooth_l1_scalar), dptr(loss_sum, f32), current_stream()))
    return grad


def se3_dist_loss_grad(rot_est_norm, rot_gt, fc7, p, zoom_trans_gt, d_rot_norm, d_zoom_trans, lw_rot, lw_trans, trans_loss_type="L2",
                       smooth_l1_scalar=3.0, loss_sums2=None):
    """SE3_DIST_LOSS (deepIM_flownet.py:396-437): adds the rot / trans loss gradients to d_rot_norm (B,4) / d_zoom_trans (B,3)"""
    check(lib().dim_se3_dist_loss_grad(dptr(rot_est_norm, f32), dptr(rot_gt, f32), dptr(fc7, f32), dptr(p["trans_weight"], f32),
                                       dptr(p["trans_bias"], f32), dptr(zoom_trans_gt, f32), dptr(d_rot_norm, f32), dptr(d_zoom_trans, f32),
                                       rot_est_norm.shape[0], float(lw_rot), float(lw_trans), LOSS_TYPE_ID[trans_loss_type],
                                       float(smooth_l1_scalar), dptr(loss_sums2, f32), current_stream()))


def quat_normalize(rot, out=None):
    out = out if out is not None else torch.empty_like(rot)
    check(lib().dim_quat_normalize(dptr(rot, f32), dptr(out, f32), rot.shape[0], current_stream()))
    return out


def pose_head_bwd(fc6a, fc7, rot_raw, d_rot_norm, d_trans, p, d_rot, dz7, dz6):
    check(lib().dim_pose_head_bwd(dptr(fc6a, f32), dptr(fc7, f32), dptr(rot_raw, f32), dptr(d_rot_norm, f32), dptr(d_trans, f32),
                                  dptr(p["fc7_weight"], f32), dptr(p["rot_weight"], f32), dptr(p["trans_weight"], f32), dptr(d_rot, f32),
                                  dptr(dz7, f32), dptr(dz6, f32), fc6a.shape[0], current_stream()))


def fc_wgrad(dz, x, dW, db=None):
    check(lib().dim_fc_wgrad(dptr(dz, f32), dptr(x, f32), dptr(dW, f32), dptr(db, f32), dz.shape[0], dz.shape[1], x.shape[1], current_stream()))


def upsample16_bwd(dout_nchw, w_c1_32_32, df_nhwc, crop=8, scale=1.0):
    N, C, OH, OW = dout_nchw.shape
    _, h, w, _ = df_nhwc.shape
    check(lib().dim_upsample16_bwd(dptr(dout_nchw, f32), dptr(w_c1_32_32, f32), dptr(df_nhwc, f32), N, C, h, w, OH, OW, crop, float(scale),
                                   current_stream()))
    return df_nhwc


def conv_small_cout_bwd(x, Cin, dy, w_oihw, dx, dw, db, accumulate_dx=False, pad=1, workspace=None):
    N, H, W, in_cs = x.shape
    Cout, _, KH, KW = w_oihw.shape
    need = lib().dim_conv_small_cout_bwd_workspace_floats(N, H, W, Cin, Cout, KH, KW)
    if workspace is None or workspace.numel() < need:
        workspace = _new((need,), x)
    check(lib().dim_conv_small_cout_bwd(dptr(x, f32), dptr(dy, f32), dptr(w_oihw, f32), dptr(dx, f32), dptr(dw, f32), dptr(db, f32),
                                        dptr(workspace, f32), N, H, W, Cin, in_cs, dx.shape[-1] if dx is not None else 0, Cout, KH, KW, pad,
                                        int(accumulate_dx), current_stream()))


def deconv4x4s2_tiny_bwd(x, dy, dy_coff, w_iohw, dx, dw, db, crop=1):
    N, H, W, x_cs = x.shape
    _, OH, OW, dy_cs = dy.shape
    Cin, Cout = w_iohw.shape[:2]
    check(lib().dim_deconv4x4s2_tiny_bwd(dptr(x, f32), x_cs, dptr(dy, f32), dy_cs, dy_coff, dptr(w_iohw, f32), dptr(dx, f32), dptr(dw, f32),
                                         dptr(db, f32), N, H, W, Cin, Cout, OH, OW, crop, current_stream()))


def sgd_momentum(w, grad, mom, lr, momentum, wd, rescale_grad=1.0):
    check(lib().dim_sgd_momentum(dptr(w, f32), dptr(grad, f32), dptr(mom, f32), w.numel(), float(lr), float(momentum), float(wd),
                                 float(rescale_grad), current_stream()))


def adam(w, grad, mean, var, lr_t, beta1=0.9, beta2=0.999, epsilon=1e-8, wd=0.0, rescale_grad=1.0):
    check(lib().dim_adam(dptr(w, f32), dptr(grad, f32), dptr(mean, f32), dptr(var, f32), w.numel(), float(lr_t), float(beta1), float(beta2),
                         float(epsilon), float(wd), float(rescale_grad), current_stream()))


def pose_to_KT(pose_src, pose_tgt, K, out=None):
    B = pose_src.shape[0]
    out = out if out is not None else _new((B, 3, 4), pose_src)
    keep, kp = host_f32(K, 9)
    check(lib().dim_pose_to_KT(dptr(pose_src, f32), dptr(pose_tgt, f32), kp, dptr(out, f32), B, current_stream()))
    return out
