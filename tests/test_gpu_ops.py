"""GPU parity: every C-ABI entry point vs the CPU oracle / the reference-generated golden vectors."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import native, se3 as ose3, transform3d as ot3d, zoom as ozoom  # noqa: E402

DEV = "cuda:0"


def cu(a, dtype=torch.float32):
    return torch.as_tensor(np.ascontiguousarray(a)).to(dtype).to(DEV)


@pytest.fixture(scope="module")
def ops(hip_lib):
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    from lib.hip import ops as _ops

    return _ops


# ------------------------------------------------------------------ SE(3): against the REFERENCE's outputs
@pytest.mark.parametrize("coord", ["MODEL", "CAMERA", "CAMERA_NEW", "NAIVE"])
def test_se3_compose_and_delta_vs_reference_golden(ops, golden_dir, coord):
    g = np.load(os.path.join(golden_dir, "se3_golden.npz"))
    ps, pt = cu(g["pose_src"]), cu(g["pose_tgt"])
    se3 = cu(np.concatenate([g["quat_raw"], g["trans_delta"]], axis=1))
    z3, o3 = np.zeros(3), np.ones(3)
    out64 = torch.empty(ps.shape, dtype=torch.float64, device=DEV)
    out = ops.se3_compose(ps, se3, coord, z3, o3, out_f64=out64)
    # inputs were rounded to f32 on upload: compare with the oracle on the same rounded inputs at 1e-12 ...
    for i in range(ps.shape[0]):
        ref = ose3.RT_transform(g["pose_src"][i].astype(np.float32).astype(np.float64), se3[i, :4].cpu().numpy().astype(np.float64),
                                se3[i, 4:].cpu().numpy().astype(np.float64), z3, o3, coord)
        np.testing.assert_allclose(out64[i].cpu().numpy(), ref, atol=1e-6 if coord == "NAIVE" else 1e-12)
    # ... and with the reference's own float64 result at f32 resolution
    np.testing.assert_allclose(out.cpu().numpy(), g[coord + "_compose"], atol=2e-6)
    rot, trans = ops.se3_delta(ps, pt, coord, z3, o3)
    np.testing.assert_allclose(rot.cpu().numpy(), g[coord + "_delta_q"], atol=2e-6)
    np.testing.assert_allclose(trans.cpu().numpy(), g[coord + "_delta_t"], atol=2e-6)
    assert (rot[:, 0] >= 0).all()
    # rot_type "MATRIX" (the reference's default): the residual rotation itself, same translation residual
    rmat, trans_m = ops.se3_delta_matrix(ps, pt, coord, z3, o3)
    np.testing.assert_allclose(rmat.cpu().numpy(), g[coord + "_delta_R"], atol=2e-6)
    np.testing.assert_array_equal(trans_m.cpu().numpy(), trans.cpu().numpy())
    from lib.pair_matching.RT_transform import calc_RT_delta

    r1, t1 = calc_RT_delta(g["pose_src"][3], g["pose_tgt"][3], z3, o3, coord, "MATRIX")
    assert r1.shape == (3, 3)
    np.testing.assert_allclose(r1, g[coord + "_delta_R"][3], atol=2e-6)
    np.testing.assert_allclose(t1, g[coord + "_delta_t"][3], atol=2e-6)
    # rot_type "EULER" and 3-number rotation deltas (RT_transform.py:39-40, :139-140), against the reference's outputs
    from lib.pair_matching.RT_transform import RT_transform

    e = np.load(os.path.join(golden_dir, "se3_euler_golden.npz"))
    se3e = cu(np.concatenate([e["euler"], g["trans_delta"]], axis=1))
    oute = ops.se3_compose_euler(ps, se3e, coord, z3, o3, out_f64=out64)
    np.testing.assert_allclose(oute.cpu().numpy(), e[coord + "_compose"], atol=2e-6)
    for i in (0, 7):   # on the f32-rounded inputs the float64 path agrees with the oracle to 1e-12
        ref = ose3.RT_transform(g["pose_src"][i].astype(np.float32).astype(np.float64), se3e[i, :3].cpu().numpy().astype(np.float64),
                                se3e[i, 3:].cpu().numpy().astype(np.float64), z3, o3, coord)
        np.testing.assert_allclose(out64[i].cpu().numpy(), ref, atol=1e-6 if coord == "NAIVE" else 1e-12)
    reul, trans_e = ops.se3_delta_euler(ps, pt, coord, z3, o3)
    np.testing.assert_allclose(reul.cpu().numpy(), e[coord + "_delta_e"], atol=4e-6)
    np.testing.assert_array_equal(trans_e.cpu().numpy(), trans.cpu().numpy())
    r3, t3 = calc_RT_delta(g["pose_src"][3], g["pose_tgt"][3], z3, o3, coord, "EULER")
    np.testing.assert_allclose(r3, e[coord + "_delta_e"][3], atol=4e-6)
    np.testing.assert_allclose(RT_transform(g["pose_src"][3], e["euler"][3], g["trans_delta"][3], z3, o3, coord), e[coord + "_compose"][3],
                               atol=2e-6)
    with pytest.raises(Exception, match="Unknown rot_type"):
        calc_RT_delta(g["pose_src"][3], g["pose_tgt"][3], z3, o3, coord, "AXIS_ANGLE")


def test_se3_means_stds(ops, golden_dir):
    g = np.load(os.path.join(golden_dir, "se3_golden.npz"))
    se3 = cu(np.concatenate([g["quat_raw"], g["trans_delta"]], axis=1))
    out = ops.se3_compose(cu(g["pose_src"]), se3, "CAMERA", g["T_means2"], g["T_stds2"])
    np.testing.assert_allclose(out.cpu().numpy(), g["ms_CAMERA_compose"], atol=2e-6)
    rot, trans = ops.se3_delta(cu(g["pose_src"]), cu(g["pose_tgt"]), "CAMERA", g["T_means2"], g["T_stds2"])
    np.testing.assert_allclose(trans.cpu().numpy(), g["ms_CAMERA_delta_t"], atol=2e-6)


def test_zoom_trans_roundtrip(ops):
    """reference check zoom_trans.py:126-168: zoom = divide by wx, inverse round-trips."""
    rng = np.random.RandomState(0)
    zf = rng.rand(8, 4).astype(np.float32) + 0.2
    zf[:, 1] = zf[:, 0]
    t = rng.randn(8, 3).astype(np.float32)
    z = ops.zoom_trans(cu(zf), cu(t), 1)
    np.testing.assert_allclose(z.cpu().numpy(), ozoom.zoom_trans(zf, t, False), rtol=1e-6)
    back = ops.zoom_trans(cu(zf), z, 2)
    np.testing.assert_allclose(back.cpu().numpy(), t, rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(ops.zoom_trans(cu(zf), cu(t), 2).cpu().numpy(), ozoom.zoom_trans(zf, t, True), rtol=1e-6)


# ------------------------------------------------------------------ Transform3D
@pytest.mark.parametrize("coord", ["MODEL", "CAMERA", "CAMERA_NEW", "NAIVE"])
def test_transform3d(ops, coord):
    from test_oracle_transform3d import _inputs

    pts, rot, trans, pose_src = _inputs()
    z3, o3 = np.zeros(3, np.float32), np.ones(3, np.float32)
    out = ops.transform3d_fwd(cu(pts), cu(rot), cu(trans), cu(pose_src), coord, z3, o3)
    ref = ot3d.forward(pts, rot, trans, pose_src, z3, o3, coord)
    np.testing.assert_allclose(out.cpu().numpy(), ref, atol=1e-5)
    # reference self-check (transform3d.py:473-493): |fwd - RT_transform| < 1e-4
    for b in range(pts.shape[0]):
        pose = ose3.RT_transform(pose_src[b], rot[b], trans[b], z3, o3, coord)
        assert np.abs(out[b].cpu().numpy() - (pose[:, :3] @ pts[b] + pose[:, 3:4])).max() < 1e-4
    g = np.random.RandomState(3).randn(*pts.shape).astype(np.float32) / pts.shape[2]
    d_rot, d_trans = ops.transform3d_bwd(cu(g), cu(pts), cu(rot), cu(trans), cu(pose_src), coord, z3, o3)
    r_rot, r_trans = ot3d.backward(g, pts, rot, trans, pose_src, z3, o3, coord)
    np.testing.assert_allclose(d_trans.cpu().numpy(), r_trans, atol=2e-5, rtol=1e-4)
    np.testing.assert_allclose(d_rot.cpu().numpy(), r_rot, atol=2e-5, rtol=1e-4)


def test_transform3d_bad_quaternion_branches(ops):
    """quat2mat_forward returns identity when |q|^2 is off by > 1e-2; backward returns 0 when off by > 1e-4."""
    from test_oracle_transform3d import _inputs

    pts, rot, trans, pose_src = _inputs()
    rot = rot * np.float32(1.2)
    z3, o3 = np.zeros(3, np.float32), np.ones(3, np.float32)
    out = ops.transform3d_fwd(cu(pts), cu(rot), cu(trans), cu(pose_src), "CAMERA", z3, o3)
    np.testing.assert_allclose(out.cpu().numpy(), ot3d.forward(pts, rot, trans, pose_src, z3, o3, "CAMERA"), atol=1e-5)
    g = np.ones_like(pts)
    d_rot, _ = ops.transform3d_bwd(cu(g), cu(pts), cu(rot), cu(trans), cu(pose_src), "CAMERA", z3, o3)
    assert (d_rot == 0).all()


# ------------------------------------------------------------------ depth -> flow
def test_depth_to_flow_vs_oracle_and_reference(ops, golden_dir):
    g = np.load(os.path.join(golden_dir, "flow_golden.npz"))
    K = g["K"]
    Kinv = np.linalg.inv(K).astype(np.float32)
    n = len(g["depth_src"])
    KT = np.zeros((n, 3, 4), dtype=np.float32)
    for i in range(n):
        R, t = ose3.calc_se3(g["pose_src"][i], g["pose_tgt"][i])
        KT[i] = np.dot(K, np.concatenate([R, t.reshape(3, 1)], axis=1)).astype(np.float32)
    ds, dt = g["depth_src"][:, None], g["depth_tgt"][:, None]
    flow, valid = ops.depth_to_flow(cu(ds), cu(dt), cu(KT), Kinv)
    rflow, rvalid = native.gpu_flow(ds, dt, KT, Kinv)
    flow, valid = flow.cpu().numpy(), valid.cpu().numpy()
    # FMA contraction may move a value by an ulp across a predicate: allow a handful of flips
    flips = (valid != rvalid).sum()
    assert flips <= 20, flips
    same = valid == rvalid
    np.testing.assert_allclose(flow[np.repeat(same, 2, axis=1)], rflow[np.repeat(same, 2, axis=1)], atol=1e-3)
    # against the reference's numpy calc_flow where both say visible
    for i in range(n):
        both = (valid[i, 0] == 1) & (g["visible"][i] == 1)
        f = np.stack([flow[i, 0], flow[i, 1]], axis=-1)
        np.testing.assert_allclose(f[both], g["flow"][i][both], atol=2e-3)
        assert both.sum() > 100


def test_depth_to_flow_quad_and_pixel_kernels_agree(ops, golden_dir):
    """dim_depth_to_flow takes four pixels per thread when W % 4 == 0 and the planes are 16-byte aligned, one pixel per thread otherwise:
    the same bits either way (a source plane placed 4 bytes off alignment forces the one-pixel form), and at 480 x 640"""
    g = np.load(os.path.join(golden_dir, "flow_golden.npz"))
    K = g["K"]
    Kinv = np.linalg.inv(K).astype(np.float32)
    n = len(g["depth_src"])
    KT = np.zeros((n, 3, 4), dtype=np.float32)
    for i in range(n):
        R, t = ose3.calc_se3(g["pose_src"][i], g["pose_tgt"][i])
        KT[i] = np.dot(K, np.concatenate([R, t.reshape(3, 1)], axis=1)).astype(np.float32)
    ds, dt = cu(g["depth_src"][:, None]), cu(g["depth_tgt"][:, None])
    f_quad, v_quad = ops.depth_to_flow(ds, dt, cu(KT), Kinv)
    buf = torch.zeros(ds.numel() + 1, device=DEV)
    ds_off = buf[1:].view(ds.shape)
    ds_off.copy_(ds)
    assert ds_off.data_ptr() % 16 == 4
    f_pix, v_pix = ops.depth_to_flow(ds_off, dt, cu(KT), Kinv)
    assert torch.equal(f_quad, f_pix) and torch.equal(v_quad, v_pix) and v_quad.sum() > 1000
    # full size: a plane of the batch tiled up to 480 x 640 (the quad kernel's h / w decode at the real width)
    big_s = ds[:2].repeat(1, 1, 4, 4).contiguous()
    big_t = dt[:2].repeat(1, 1, 4, 4).contiguous()
    assert big_s.shape[2:] == (480, 640)
    fq, vq = ops.depth_to_flow(big_s, big_t, cu(KT[:2]), Kinv)
    buf = torch.zeros(big_s.numel() + 1, device=DEV)
    off = buf[1:].view(big_s.shape)
    off.copy_(big_s)
    fp, vp = ops.depth_to_flow(off, big_t, cu(KT[:2]), Kinv)
    assert torch.equal(fq, fp) and torch.equal(vq, vp)


def test_depth_to_flow_empty_and_zero_depth(ops):
    z = torch.zeros((2, 1, 48, 64), device=DEV)
    KT = cu(np.tile(np.eye(3, 4, dtype=np.float32), (2, 1, 1)))
    flow, valid = ops.depth_to_flow(z, z, KT, np.eye(3, dtype=np.float32))
    assert flow.abs().sum() == 0 and valid.sum() == 0
    flow, valid = ops.depth_to_flow(z[:0], z[:0], KT[:0], np.eye(3, dtype=np.float32))
    assert flow.shape[0] == 0


# ------------------------------------------------------------------ zoom ops
def _rand_masks(rng, B, H, W, empty_rendered=()):
    mo = np.zeros((B, 1, H, W), np.float32)
    mr = np.zeros((B, 1, H, W), np.float32)
    pose = np.zeros((B, 3, 4), np.float32)
    for b in range(B):
        y0, x0 = rng.randint(60, 200), rng.randint(80, 300)
        h, w = rng.randint(40, 200), rng.randint(40, 250)
        mo[b, 0, y0:y0 + h, x0:x0 + w] = 1
        if b not in empty_rendered:
            yy, xx = np.mgrid[0:H, 0:W]
            cy, cx = y0 + h / 2 + rng.randint(-15, 15), x0 + w / 2 + rng.randint(-15, 15)
            mr[b, 0] = (((yy - cy) / (h / 2.2)) ** 2 + ((xx - cx) / (w / 2.2)) ** 2 < 1) * rng.uniform(0.3, 1.5)  # depth-like values
        z = rng.uniform(0.5, 1.2)
        pose[b] = np.concatenate([np.eye(3), [[(x0 + w / 2 - 325.26) * z / 572.4], [(y0 + h / 2 - 242.05) * z / 573.6], [z]]], axis=1)
    return mo, mr, pose


def test_zoom_mask_and_image_vs_oracle(ops):
    from lib.utils.synthetic import LINEMOD_K, PIXEL_MEANS

    rng = np.random.RandomState(5)
    B, H, W = 4, 480, 640
    mo, mr, pose = _rand_masks(rng, B, H, W, empty_rendered=(3,))
    io = (rng.randint(0, 256, size=(B, 3, H, W)) - PIXEL_MEANS[::-1].reshape(1, 3, 1, 1)).astype(np.float32)
    ir = (rng.randint(0, 256, size=(B, 3, H, W)) - PIXEL_MEANS[::-1].reshape(1, 3, 1, 1)).astype(np.float32)
    zmo, zmg, zmr, zf = ozoom.zoom_mask(mo, mo, mr, pose, LINEMOD_K, H, W)
    zio, zir = ozoom.zoom_image_with_factor(zf, io, ir, PIXEL_MEANS, H, W)
    bo = ops.mask_bbox(cu(mo), 0.3)
    br = ops.mask_bbox(cu(mr), 0.2)
    assert br[3].tolist() == [W, -1, H, -1]  # empty rendered mask
    status = torch.zeros(B, dtype=torch.int32, device=DEV)
    gzf = ops.zoom_factor(bo, br, cu(pose), LINEMOD_K, H, W, status=status)
    np.testing.assert_allclose(gzf.cpu().numpy(), zf, rtol=2e-6, atol=2e-6)
    assert status.tolist() == [0, 0, 0, 2]
    # use the oracle's factor so sampling differences are not confounded with factor rounding
    zft = cu(zf)
    pm = PIXEL_MEANS[::-1].copy()
    nchw = tuple(torch.empty_like(t) for t in (cu(io), cu(ir), cu(mo), cu(mr)))
    X = ops.zoom_net_input(cu(io), cu(ir), cu(mo), cu(mr), zft, pm, nchw_out=nchw)
    np.testing.assert_allclose(nchw[0].cpu().numpy(), zio, atol=2e-3)
    np.testing.assert_allclose(nchw[1].cpu().numpy(), zir, atol=2e-3)
    assert (nchw[2].cpu().numpy() != zmo).sum() <= 8
    assert (nchw[3].cpu().numpy() != zmr).sum() <= 8
    Xr = np.concatenate([zio / 255.0, zir / 255.0, zmo, zmr], axis=1).transpose(0, 2, 3, 1)
    Xg = X.cpu().numpy()
    np.testing.assert_allclose(Xg[..., :6], Xr[..., :6], atol=1e-5)
    assert (Xg[..., 6:] != Xr[..., 6:]).sum() <= 16
    # generic plane sampler: ZoomMaskWithFactor fwd / inverse, ZoomDepth, ZoomFlow fwd / inverse
    m = ops.zoom_planes(cu(mr), zft, pre=1, post=1)
    assert (m.cpu().numpy() != ozoom.zoom_mask_with_factor(zf, mr, False)).sum() <= 8
    mi = ops.zoom_planes(cu(zmr), zft, inverse=True, pre=1, post=1)
    assert (mi.cpu().numpy() != ozoom.zoom_mask_with_factor(zf, zmr, True)).sum() <= 8
    d = ops.zoom_planes(cu(mr), zft)
    np.testing.assert_allclose(d.cpu().numpy(), ozoom.zoom_depth(zf, mr, mr)[0], atol=1e-5)
    flow = rng.randn(B, 2, H, W).astype(np.float32) * 5
    fw = (rng.rand(B, 2, H, W) > 0.5).astype(np.float32)
    rf, rw = ozoom.zoom_flow(zf, flow, fw, b_inv_zoom=False)
    gf = ops.zoom_planes(cu(flow), zft, scale_mode=1)
    gw = ops.zoom_planes(cu(fw), zft, post=2)
    np.testing.assert_allclose(gf.cpu().numpy(), rf, atol=1e-3, rtol=1e-5)
    assert (gw.cpu().numpy() != rw).sum() <= 16
    gfi = ops.zoom_planes(cu(rf), zft, inverse=True, scale_mode=2)
    np.testing.assert_allclose(gfi.cpu().numpy(), ozoom.zoom_flow(zf, rf, b_inv_zoom=True), atol=1e-3, rtol=1e-5)


def test_zoom_image_mode_bbox(ops):
    """ZoomImage validity rule (zoom_image.py:31-37): sum_c(image + mean) > 0.01."""
    from lib.utils.synthetic import LINEMOD_K, PIXEL_MEANS

    rng = np.random.RandomState(9)
    B, H, W = 2, 480, 640
    pm = PIXEL_MEANS[::-1].reshape(1, 3, 1, 1)
    raw_o = np.zeros((B, 3, H, W), np.float32)
    raw_r = np.zeros((B, 3, H, W), np.float32)
    raw_o[:, :, 100:300, 200:420] = rng.randint(1, 255, size=(B, 3, 200, 220))
    raw_r[:, :, 120:310, 180:400] = rng.randint(1, 255, size=(B, 3, 190, 220))
    io, ir = (raw_o - pm).astype(np.float32), (raw_r - pm).astype(np.float32)
    pose = np.tile(np.array([[1, 0, 0, -0.03], [0, 1, 0, -0.04], [0, 0, 1, 0.9]], np.float32), (B, 1, 1))
    zio, zir, zf = ozoom.zoom_image(io, ir, pose, LINEMOD_K, PIXEL_MEANS, H, W)
    bo = ops.mask_bbox(cu(io), 0.01, mode=1, means3=pm.reshape(3))
    br = ops.mask_bbox(cu(ir), 0.01, mode=1, means3=pm.reshape(3))
    gzf = ops.zoom_factor(bo, br, cu(pose), LINEMOD_K, H, W)
    np.testing.assert_allclose(gzf.cpu().numpy(), zf, rtol=2e-6, atol=2e-6)
    out = ops.zoom_planes(cu(io), cu(zf), add3=pm.reshape(3))
    np.testing.assert_allclose(out.cpu().numpy(), zio, atol=2e-3)


def test_zoom_observed_empty_sets_status(ops):
    from lib.utils.synthetic import LINEMOD_K

    B, H, W = 2, 480, 640
    mo = torch.zeros((B, 1, H, W), device=DEV)
    mr = torch.zeros((B, 1, H, W), device=DEV)
    mr[1, 0, 100:200, 100:200] = 1
    pose = cu(np.tile(np.eye(3, 4, dtype=np.float32), (B, 1, 1)) + np.array([0, 0, 0, 1], np.float32))
    status = torch.zeros(B, dtype=torch.int32, device=DEV)
    zf = ops.zoom_factor(ops.mask_bbox(mo, 0.3), ops.mask_bbox(mr, 0.2), pose, LINEMOD_K, H, W, status=status)
    assert status.tolist() == [3, 1]  # reference raises ValueError here (np.min of empty)
    assert torch.isfinite(zf).all()


# ------------------------------------------------------------------ rasteriser
def test_rasteriser_vs_oracle(ops):
    from lib.render_hip.render_py_multi import Render_Py
    from lib.utils import synthetic as syn

    models = syn.make_models(seed=3, n_models=2, subdiv=3)
    cls, gt, init = syn.sample_pairs(4, 4, n_classes=2)
    K = syn.LINEMOD_K
    for bil in (False, True):
        rm = Render_Py(None, ["a", "b"], K, meshes=models, tex_bilinear=bil)
        B = 4
        image = torch.empty((B, 3, 480, 640), device=DEV)
        depth = torch.empty((B, 1, 480, 640), device=DEV)
        mask = torch.empty((B, 1, 480, 640), device=DEV)
        bgr = torch.empty((B, 480, 640, 3), device=DEV)
        bbox = torch.empty((B, 4), dtype=torch.int32, device=DEV)
        pm = syn.plane_means()
        rm.render_batch(cu(cls, torch.int32), cu(init), image=image, depth=depth, mask=mask, bgr=bgr, bbox=bbox, plane_means=pm)
        for b in range(B):
            v, t, f, tex = models[cls[b]]
            rb, rd = native.render(v, t, f, tex, init[b][:, :3], init[b][:, 3], K, tex_bilinear=bil)
            gd = depth[b, 0].cpu().numpy()
            cov_diff = ((gd > 0) != (rd > 0)).sum()
            assert cov_diff <= 4, cov_diff
            both = (gd > 0) & (rd > 0)
            assert both.sum() > 500
            np.testing.assert_allclose(gd[both], rd[both], rtol=2e-6)
            gb = bgr[b].cpu().numpy()
            bad = (np.abs(gb - rb).max(axis=-1) > (1.0 if bil else 0.0)) & both
            assert bad.sum() <= (20 if bil else 8), bad.sum()  # texel flips on cell borders
            np.testing.assert_array_equal(mask[b, 0].cpu().numpy(), (gd > 0.2).astype(np.float32))
            ys, xs = np.nonzero(gd > 0.2)
            assert bbox[b].tolist() == [xs.min(), xs.max(), ys.min(), ys.max()]
            ref_img = syn.bgr_to_blob(gb)[0]
            np.testing.assert_allclose(image[b].cpu().numpy(), ref_img, atol=1e-4)
        # reference object API: render(cls_idx, r, t, r_type='mat')
        b0, d0 = rm.render(int(cls[0]), init[0][:, :3], init[0][:, 3], r_type="mat")
        np.testing.assert_array_equal(d0, depth[0, 0].cpu().numpy())
        np.testing.assert_array_equal(b0, bgr[0].cpu().numpy())


def test_rasteriser_offscreen_and_clipping(ops):
    from lib.render_hip.render_py_multi import Render_Py
    from lib.utils import synthetic as syn

    models = syn.make_models(seed=3, n_models=1, subdiv=2)
    rm = Render_Py(None, ["a"], syn.LINEMOD_K, meshes=models)
    poses = np.tile(np.eye(3, 4, dtype=np.float32), (3, 1, 1))
    poses[0, :, 3] = [5.0, 0, 1.0]    # far off-screen
    poses[1, :, 3] = [0, 0, 0.1]      # closer than znear: every fragment clipped
    poses[2, :, 3] = [0, 0, -1.0]     # behind the camera
    depth = torch.empty((3, 1, 480, 640), device=DEV)
    bbox = torch.empty((3, 4), dtype=torch.int32, device=DEV)
    mask = torch.empty((3, 1, 480, 640), device=DEV)
    rm.render_batch(cu(np.zeros(3), torch.int32), cu(poses), depth=depth, mask=mask, bbox=bbox)
    assert depth.abs().sum() == 0
    assert bbox.tolist() == [[640, -1, 480, -1]] * 3
    m = torch.ones((3, 1, 480, 640), device=DEV)
    ops.box_mask(bbox, m)
    assert m.sum() == 0


def test_rasteriser_near_plane_clipping_vs_oracle(ops):
    """GL clips against zNear = 0.25 (render_py_multi.py:152-169): meshes that straddle the plane -- and one with vertices BEHIND the
    eye -- against oracle/raster.c (itself checked against ray casting: tests/test_oracle_crosscheck.py), unlit and lit.  Nothing
    in front of the plane, no hole where a cut triangle's remainder shows, colours from camera-space barycentrics of the original
    triangle."""
    from lib.render_hip.render_py_light_modelnet_multi import Render_Py_Light_ModelNet_Multi, vertex_normals
    from lib.render_hip.render_py_multi import Render_Py
    from lib.utils import synthetic as syn

    models = syn.make_models(seed=11, n_models=2, subdiv=3)
    K = syn.LINEMOD_K
    rng = np.random.default_rng(8)
    B = 4
    cls = np.array([0, 1, 0, 1], np.int32)
    big = []   # 0.6 m across: deep enough along any view direction to reach from behind the eye to beyond the near plane
    for v, t, f, tex in models:
        v = v * (0.6 / (v.max(0) - v.min(0)).max())
        big.append((v.astype(np.float32), t, f, tex))
    poses = np.zeros((B, 3, 4), np.float32)
    for b in range(B):
        q = rng.normal(size=4)
        R = ose3.quat2mat(q / np.linalg.norm(q))
        zr = (big[cls[b]][0] @ R.T)[:, 2]
        # near plane through the middle of the object / only the far third beyond it / the last two also reach behind the eye (Z < 0)
        tz = (0.25 - 0.5 * (zr.min() + zr.max()), 0.25 - 0.67 * zr.max(), -0.5 * zr.min(), -0.8 * zr.min())[b]
        poses[b, :, :3], poses[b, :, 3] = R, [0.01 * b, -0.01, tz]
        zc = zr + tz
        assert zc.min() < 0.25 < zc.max() and (b < 2 or zc.min() < -0.02), (b, zc.min(), zc.max())
    for bil in (False, True):
        rm = Render_Py(None, ["a", "b"], K, meshes=big, tex_bilinear=bil)
        depth = torch.empty((B, 1, 480, 640), device=DEV)
        bgr = torch.empty((B, 480, 640, 3), device=DEV)
        status = torch.zeros(B, dtype=torch.int32, device=DEV)
        rm.render_batch(cu(cls, torch.int32), cu(poses), depth=depth, bgr=bgr, status=status)
        assert status.tolist() == [0] * B
        for b in range(B):
            v, t, f, tex = big[cls[b]]
            rb, rd = native.render(v, t, f, tex, poses[b][:, :3], poses[b][:, 3], K, tex_bilinear=bil)
            gd, gb = depth[b, 0].cpu().numpy(), bgr[b].cpu().numpy()
            assert (rd > 0).sum() > 3000 and rd[rd > 0].min() >= 0.25 and gd[gd > 0].min() >= 0.25
            assert ((gd > 0) != (rd > 0)).sum() <= 4
            both = (gd > 0) & (rd > 0)
            np.testing.assert_allclose(gd[both], rd[both], rtol=2e-6)
            bad = (np.abs(gb - rb).max(axis=-1) > (1.0 if bil else 0.0)) & both
            assert bad.sum() <= (20 if bil else 8), (b, bil, bad.sum())
    # lit variant: normals / positions of a clipped face come from the same camera-space barycentrics
    gray = np.full((32, 32, 3), 180, np.uint8)
    meshes = [(v, vertex_normals(v, f).astype(np.float32), t, f) for v, t, f, _ in big]
    rml = Render_Py_Light_ModelNet_Multi(None, gray, K, 640, 480, 0.25, 6.0, brightness_ratios=[0.7], meshes=meshes)
    depth = torch.empty((B, 1, 480, 640), device=DEV)
    bgr = torch.empty((B, 480, 640, 3), device=DEV)
    li = np.tile(np.array([1.0, 0.95, 1.05], np.float32), (B, 1))
    lp = np.stack([native.modelnet_light_position(poses[b].astype(np.float64), idx=2) for b in range(B)]).astype(np.float32)
    rml.render_batch(cu(cls, torch.int32), cu(poses), depth=depth, bgr=bgr, light_position=cu(lp), light_intensity=cu(li))
    for b in range(B):
        v, n, t, f = meshes[cls[b]]
        rb, rd = native.render_lit(v, n, t, f, gray, poses[b][:, :3], poses[b][:, 3], K, lp[b], li[b], 0.7)
        gd, gb = depth[b, 0].cpu().numpy(), bgr[b].cpu().numpy()
        assert ((gd > 0) != (rd > 0)).sum() <= 4
        both = (gd > 0) & (rd > 0)
        assert (np.abs(gb - rb).max(axis=-1)[both] > 1.0).sum() <= 8


def test_rasteriser_reports_a_class_index_outside_the_mesh_table(ops):
    """class_index is data: one outside [0, n_classes) must neither read past mesh_table nor pass silently -- the sample renders as
    background and DIM_STATUS_BAD_CLASS (4) is OR-ed into its status word; the other samples are unaffected."""
    from lib.render_hip.render_py_multi import Render_Py
    from lib.utils import synthetic as syn

    models = syn.make_models(seed=3, n_models=2, subdiv=2)
    rm = Render_Py(None, ["a", "b"], syn.LINEMOD_K, meshes=models)
    cls, gt, init = syn.sample_pairs(4, 3, n_classes=2)
    bad = cls.copy()
    bad[1] = 7
    out = {}
    for tag, c in (("good", cls), ("bad", bad)):
        depth = torch.empty((3, 1, 480, 640), device=DEV)
        image = torch.empty((3, 3, 480, 640), device=DEV)
        bbox = torch.empty((3, 4), dtype=torch.int32, device=DEV)
        status = torch.tensor([0, 1, 0], dtype=torch.int32, device=DEV)  # bits already set by another kernel survive
        rm.render_batch(cu(c, torch.int32), cu(init), image=image, depth=depth, bbox=bbox, plane_means=syn.plane_means(), status=status)
        out[tag] = (depth.cpu().numpy(), image.cpu().numpy(), bbox.tolist(), status.tolist())
    assert out["good"][3] == [0, 1, 0] and out["bad"][3] == [0, 1 | 4, 0]
    assert out["bad"][0][1].sum() == 0 and out["bad"][2][1] == [640, -1, 480, -1]
    np.testing.assert_array_equal(out["bad"][1][1], np.broadcast_to(-syn.plane_means().reshape(3, 1, 1), (3, 480, 640)))
    for b in (0, 2):
        np.testing.assert_array_equal(out["bad"][0][b], out["good"][0][b])
        np.testing.assert_array_equal(out["bad"][1][b], out["good"][1][b])
    with pytest.raises(Exception):  # negative / zero table sizes are argument errors on the host
        ops.check(ops.lib().dim_raster_render(None, None, None, None, 0, 1, 1, None, None, None, None, None, 1, 480, 640, 0.25, 6.0, 0, None, 0.2,
                                              None, None, None, None, None, None, None, None))


def test_box_mask_end_exclusive(ops):
    bbox = torch.tensor([[10, 20, 30, 50], [5, 5, 7, 9]], dtype=torch.int32, device=DEV)
    m = torch.empty((2, 1, 480, 640), device=DEV)
    ops.box_mask(bbox, m)
    ref = np.zeros((2, 1, 480, 640), np.float32)
    ref[0, 0, 30:50, 10:20] = 1  # data_pair.py:114 mask[y_start:y_end, x_start:x_end]
    np.testing.assert_array_equal(m.cpu().numpy(), ref)


def test_box_mask_reports_the_bbox_a_scan_would_find(ops):
    """dim_box_mask's bbox_of_mask output replaces the next iteration's ZoomMask scan of mask_observed: it must equal
    dim_mask_bbox of the mask just written, including empty and degenerate rectangles."""
    bbox = torch.tensor([[10, 20, 30, 50], [5, 5, 7, 9], [640, -1, 480, -1], [0, 640, 0, 480], [3, 4, 479, 480], [7, 9, 12, 12]],
                        dtype=torch.int32, device=DEV)
    m = torch.empty((6, 1, 480, 640), device=DEV)
    got = torch.full((6, 4), -7, dtype=torch.int32, device=DEV)
    ops.box_mask(bbox, m, bbox_of_mask=got)
    scan = ops.mask_bbox(m, 0.3)
    assert got.tolist() == scan.tolist()
    assert got[0].tolist() == [10, 19, 30, 49] and got[2].tolist() == [640, -1, 480, -1]


# ------------------------------------------------------------------ convolution
CONV_CASES = [
    # N, H, W, Cin, Cout, k, s, p, tile, splits
    (2, 60, 80, 8, 64, 7, 2, 3, 0, 1),
    (1, 37, 53, 8, 64, 7, 2, 3, 2, 1),
    (1, 37, 53, 8, 64, 7, 2, 3, 3, 1),
    (2, 60, 80, 8, 64, 7, 2, 3, 6, 1),     # tile 6: the LDS-halo first-layer kernel (8 x 16 output blocks; 30 x 40 = no partial block ...
    (1, 37, 53, 8, 64, 7, 2, 3, 6, 1),     # ... 19 x 27 outputs: partial blocks on both axes)
    (3, 22, 18, 8, 64, 7, 2, 3, 6, 1),     # image smaller than one patch
    (2, 30, 40, 64, 128, 5, 2, 2, 1, 1),
    (1, 23, 31, 64, 128, 5, 2, 2, 3, 1),
    (2, 15, 20, 256, 256, 3, 1, 1, 0, 1),
    (2, 15, 20, 256, 512, 3, 2, 1, 1, 1),
    (2, 15, 20, 512, 512, 3, 1, 1, 3, 3),
    (2, 8, 10, 512, 1024, 3, 2, 1, 0, 4),
    (1, 9, 11, 32, 64, 3, 1, 0, 2, 1),
]


@pytest.mark.parametrize("case", CONV_CASES)
def test_conv2d_vs_torch_cpu(ops, case, wino_split):
    import torch.nn.functional as F

    N, H, W, Cin, Cout, k, s, p, tile, splits = case
    g = torch.Generator().manual_seed(sum(case))
    x = torch.randn((N, Cin, H, W), generator=g)
    w = torch.randn((Cout, Cin, k, k), generator=g) / np.sqrt(Cin * k * k)
    b = torch.randn((Cout,), generator=g)
    ref = F.leaky_relu(F.conv2d(x.double(), w.double(), b.double(), stride=s, padding=p), 0.1).float()
    wp = ops.conv2d_pack_weight(w.to(DEV))
    y = ops.conv2d_fwd(x.permute(0, 2, 3, 1).contiguous().to(DEV), wp, b.to(DEV), Cout, k, k, s, p, slope=0.1, splits=splits, tile=tile)
    got = y.permute(0, 3, 1, 2).cpu()
    assert got.shape == ref.shape
    np.testing.assert_allclose(got.numpy(), ref.numpy(), atol=2e-5, rtol=1e-4)


def test_conv2d_linear_no_bias_and_arg_errors(ops):
    import torch.nn.functional as F
    from lib.hip.capi import DeepIMHipError

    x = torch.randn((1, 32, 9, 9))
    w = torch.randn((64, 32, 3, 3)) / 17
    wp = ops.conv2d_pack_weight(w.to(DEV))
    y = ops.conv2d_fwd(x.permute(0, 2, 3, 1).contiguous().to(DEV), wp, None, 64, 3, 3, 1, 1, slope=1.0)
    np.testing.assert_allclose(y.permute(0, 3, 1, 2).cpu().numpy(), F.conv2d(x, w, padding=1).numpy(), atol=2e-5, rtol=1e-4)
    with pytest.raises(DeepIMHipError):
        ops.conv2d_fwd(torch.zeros((1, 9, 9, 12), device=DEV), wp, None, 64, 3, 3, 1, 1)
    with pytest.raises(DeepIMHipError):
        ops.conv2d_fwd(torch.zeros((1, 9, 9, 32)), wp, None, 64, 3, 3, 1, 1)  # CPU tensor: no fallback


def test_fc6_and_pose_head(ops):
    import torch.nn.functional as F

    g = torch.Generator().manual_seed(1)
    B = 3
    feat = torch.randn((B, 1024, 8, 10), generator=g)
    w6 = torch.randn((256, 81920), generator=g) / 286
    b6 = torch.randn((256,), generator=g)
    p = {"fc7_weight": torch.randn((256, 256), generator=g) / 16, "fc7_bias": torch.randn((256,), generator=g),
         "rot_weight": torch.randn((4, 256), generator=g) / 16, "rot_bias": torch.randn((4,), generator=g),
         "trans_weight": torch.randn((3, 256), generator=g) / 16, "trans_bias": torch.randn((3,), generator=g)}
    zf = torch.rand((B, 4), generator=g) + 0.3
    fc6 = F.leaky_relu(F.linear(feat.reshape(B, -1).double(), w6.double(), b6.double()), 0.1)
    fc7 = F.leaky_relu(F.linear(fc6, p["fc7_weight"].double(), p["fc7_bias"].double()), 0.1)
    rot = F.linear(fc7, p["rot_weight"].double(), p["rot_bias"].double())
    tr = F.linear(fc7, p["trans_weight"].double(), p["trans_bias"].double())
    tr[:, :2] *= zf[:, :1].double()
    ref = torch.cat([rot, tr], dim=1).float().numpy()
    wp = ops.fc_pack_weight(w6.to(DEV), 1024, 8, 10)
    y6 = ops.conv2d_fwd(feat.permute(0, 2, 3, 1).contiguous().to(DEV), wp, b6.to(DEV), 256, 8, 10, 1, 0, slope=0.1, splits=160, tile=3)
    np.testing.assert_allclose(y6.view(B, 256).cpu().numpy(), fc6.float().numpy(), atol=5e-5, rtol=1e-4)
    se3 = ops.pose_head_fwd(y6.view(B, 256), {k: v.to(DEV) for k, v in p.items()}, zf.to(DEV))
    np.testing.assert_allclose(se3.cpu().numpy(), ref, atol=5e-5, rtol=1e-4)


@pytest.fixture(params=[1, 0], ids=["split3xbf16", "f32pipe"])
def wino_split(request, hip_lib):
    """both arithmetics of the Winograd plane GEMMs: f32 operands as three bf16 terms / six MFMA products (the default), and the f32 pipe"""
    from lib.hip import ops

    ops.set_winograd_split(bool(request.param))
    yield request.param
    ops.set_winograd_split(True)


@pytest.mark.parametrize("m", [2, 4])
# the last shape has more (tile, plane) items than resident workgroups: stream-K ranges that start / end inside an item
@pytest.mark.parametrize("shape", [(2, 15, 20, 64, 128), (1, 8, 10, 96, 64), (3, 30, 40, 256, 256), (2, 7, 9, 32, 64), (1, 3, 5, 32, 64),
                                   (16, 58, 80, 64, 128),
                                   # few tiles, many weights (conv6_1-like): fewer rows than one workgroup tile
                                   (2, 8, 10, 512, 512), (7, 7, 10, 256, 1024), (16, 8, 10, 256, 1024)])
def test_conv3x3_winograd_vs_f64(hip_lib, shape, m, wino_split):
    """Winograd F(2x2,3x3) / F(4x4,3x3) paths (odd and even H/W, partial edge tiles, bias + LeakyReLU) vs torch-CPU float64 conv2d"""
    import torch.nn.functional as F
    from lib.hip import ops

    N, H, W, Cin, Cout = shape
    g = torch.Generator().manual_seed(sum(shape))
    x = torch.randn((N, Cin, H, W), generator=g)
    w = torch.randn((Cout, Cin, 3, 3), generator=g) * (1.0 / np.sqrt(9 * Cin))
    b = torch.randn((Cout,), generator=g) * 0.1
    ref = F.leaky_relu(F.conv2d(x.double(), w.double(), b.double(), stride=1, padding=1), 0.1).permute(0, 2, 3, 1).numpy()
    xd = x.permute(0, 2, 3, 1).contiguous().to("cuda:0")
    wp = ops.winograd_pack_weight(w.to("cuda:0"), m=m)
    # 5 = 128 x 256 workgroup tile; 6 / 7 = 160 x 128 / 96 x 128 (the few-row layers: four waves, a 32-column strip of all rows each)
    for tile in ((3, 4, 5, 6, 7) if Cout % 256 == 0 else (3, 4, 6, 7)) if Cout % 128 == 0 else (3,):
        y = ops.conv2d_fwd_winograd(xd, Cin, wp, b.to("cuda:0"), Cout, slope=0.1, tile=tile, m=m).cpu().numpy()
        err = np.abs(y - ref).max()
        assert err <= 1e-4 * np.abs(ref).max() + 2e-5, (tile, err)
    # strided output (concat buffer) and padded input channels
    xd2 = torch.zeros((N, H, W, Cin + 32), device="cuda:0")
    xd2[..., :Cin] = xd
    out = torch.full((N, H, W, Cout + 64), 7.0, device="cuda:0")
    ops.conv2d_fwd_winograd(xd2, Cin, wp, b.to("cuda:0"), Cout, slope=0.1, tile=3, out=out, out_coff=32, m=m)
    o = out.cpu().numpy()
    assert np.abs(o[..., 32:32 + Cout] - ref).max() <= 1e-4 * np.abs(ref).max() + 2e-5
    assert np.all(o[..., :32] == 7.0) and np.all(o[..., 32 + Cout:] == 7.0)


@pytest.mark.parametrize("shape", [(2, 30, 40, 64, 128), (1, 16, 24, 32, 64), (2, 15, 21, 64, 64), (1, 6, 9, 32, 128), (3, 60, 80, 128, 256)])
def test_conv5x5s2_winograd_vs_f64(hip_lib, shape, wino_split):
    """5x5 / stride-2 / pad-2 layer (conv2, conv3) as four phase images through Winograd F(4x4,3x3): even and odd H/W, partial
    tiles, bias + LeakyReLU, strided output, padded input channels -- vs torch-CPU float64 conv2d and vs the direct MFMA kernel"""
    import torch.nn.functional as F
    from lib.hip import ops

    N, H, W, Cin, Cout = shape
    g = torch.Generator().manual_seed(sum(shape) + 5)
    x = torch.randn((N, Cin, H, W), generator=g)
    w = torch.randn((Cout, Cin, 5, 5), generator=g) * (1.0 / np.sqrt(25 * Cin))
    b = torch.randn((Cout,), generator=g) * 0.1
    ref = F.leaky_relu(F.conv2d(x.double(), w.double(), b.double(), stride=2, padding=2), 0.1).permute(0, 2, 3, 1).numpy()
    xd = x.permute(0, 2, 3, 1).contiguous().to("cuda:0")
    wp = ops.winograd5x5s2_pack_weight(w.to("cuda:0"))
    for tile in ((3, 4, 5, 6, 7) if Cout % 256 == 0 else (3, 4, 6, 7)) if Cout % 128 == 0 else (3,):
        y = ops.conv2d_fwd_winograd5x5s2(xd, Cin, wp, b.to("cuda:0"), Cout, slope=0.1, tile=tile).cpu().numpy()
        assert y.shape == ref.shape
        err = np.abs(y - ref).max()
        assert err <= 1e-4 * np.abs(ref).max() + 2e-5, (tile, err)
    yd = ops.conv2d_fwd(xd, ops.conv2d_pack_weight(w.to("cuda:0")), b.to("cuda:0"), Cout, 5, 5, 2, 2, slope=0.1, tile=3).cpu().numpy()
    assert np.abs(y - yd).max() <= 1e-4 * np.abs(ref).max() + 2e-5
    xd2 = torch.zeros((N, H, W, Cin + 32), device="cuda:0")
    xd2[..., :Cin] = xd
    out = torch.full(ref.shape[:3] + (Cout + 64,), 7.0, device="cuda:0")
    ops.conv2d_fwd_winograd5x5s2(xd2, Cin, wp, b.to("cuda:0"), Cout, slope=0.1, tile=3, out=out, out_coff=32)
    o = out.cpu().numpy()
    assert np.abs(o[..., 32:32 + Cout] - ref).max() <= 1e-4 * np.abs(ref).max() + 2e-5
    assert np.all(o[..., :32] == 7.0) and np.all(o[..., 32 + Cout:] == 7.0)


@pytest.mark.parametrize("shape", [(2, 30, 40, 64, 128), (1, 15, 21, 32, 64), (2, 60, 80, 128, 256), (1, 7, 10, 16, 32), (16, 60, 78, 32, 64)])
def test_conv5x5s2_winograd_dgrad_vs_f64(hip_lib, shape, wino_split):
    """input gradient of the 5x5 / stride-2 / pad-2 layer through Winograd (one transform of dY, K = Cout, N = 4 Cin, phase scatter)
    vs torch-CPU float64 autograd and vs the direct dgrad path; even / odd H and W; padded channel strides on both sides"""
    import torch.nn.functional as F
    from lib.hip import ops

    N, H, W, Cin, Cout = shape
    g = torch.Generator().manual_seed(sum(shape) + 13)
    w = torch.randn((Cout, Cin, 5, 5), generator=g) * (1.0 / np.sqrt(25 * Cout))
    Ho, Wo = (H + 1) // 2, (W + 1) // 2
    dy = torch.randn((N, Cout, Ho, Wo), generator=g)
    x = torch.zeros((N, Cin, H, W), dtype=torch.float64, requires_grad=True)
    F.conv2d(x, w.double(), None, stride=2, padding=2).backward(dy.double())
    ref = x.grad.permute(0, 2, 3, 1).numpy()
    dyd = torch.zeros((N, Ho, Wo, Cout + 32), device="cuda:0")
    dyd[..., :Cout] = dy.permute(0, 2, 3, 1).to("cuda:0")
    wp = ops.winograd5x5s2_dgrad_pack_weight(w.to("cuda:0"))
    dx = torch.full((N, H, W, Cin + 8), 7.0, device="cuda:0")
    ops.conv2d_dgrad_winograd5x5s2(dyd, Cout, wp, dx, Cin)
    got = dx.cpu().numpy()
    tol = 1e-4 * np.abs(ref).max() + 2e-5
    assert np.abs(got[..., :Cin] - ref).max() <= tol
    assert np.all(got[..., Cin:] == 7.0)
    if Cin % 64:
        return  # the direct dgrad path wants dx channels in multiples of 64
    dx2 = torch.empty((N, H, W, Cin), device="cuda:0")
    ops.conv2d_dgrad(dyd[..., :Cout].contiguous(), Cout, ops.conv2d_dgrad_pack_weight(w.to("cuda:0"), 2, 2), dx2, Cin, 5, 5, 2, 2, accumulate=False)
    assert np.abs(got[..., :Cin] - dx2.cpu().numpy()).max() <= tol


@pytest.mark.parametrize("shape", [(2, 30, 40, 64, 128, 1), (1, 15, 21, 64, 64, 1), (3, 13, 18, 128, 256, 1), (2, 30, 40, 32, 64, 2),
                                   (1, 15, 21, 64, 128, 2), (4, 24, 34, 32, 64, 2)])
def test_conv_winograd_wgrad_vs_f64(hip_lib, shape):
    """weight gradient through Winograd (F(3x3,4x4) per dY tile; S = 1: 3x3 / stride 1, S = 2: 5x5 / stride 2 over phase images) vs
    torch-CPU float64 autograd; padded channel strides, several pixel splits, scale / accumulate"""
    import torch.nn.functional as F
    from lib.hip import ops

    N, H, W, Cin, Cout, S = shape
    k, pad = (3, 1) if S == 1 else (5, 2)
    g = torch.Generator().manual_seed(sum(shape) + 17)
    x = torch.randn((N, Cin, H, W), generator=g)
    Ho, Wo = (H, W) if S == 1 else ((H + 1) // 2, (W + 1) // 2)
    dy = torch.randn((N, Cout, Ho, Wo), generator=g) / np.sqrt(N * Ho * Wo)
    w = torch.zeros((Cout, Cin, k, k), dtype=torch.float64, requires_grad=True)
    F.conv2d(x.double(), w, None, stride=S, padding=pad).backward(dy.double())
    ref = w.grad.numpy()
    xd = torch.zeros((N, H, W, Cin + 32), device="cuda:0")
    xd[..., :Cin] = x.permute(0, 2, 3, 1).to("cuda:0")
    dyd = torch.zeros((N, Ho, Wo, Cout + 64), device="cuda:0")
    dyd[..., :Cout] = dy.permute(0, 2, 3, 1).to("cuda:0")
    tol = 1e-4 * np.abs(ref).max() + 2e-6
    for splits in (1, 3):
        dw = torch.full((Cout, Cin, k, k), 9.0, device="cuda:0")
        ops.conv2d_wgrad_winograd(xd, Cin, dyd, Cout, dw, S=S, splits=splits)
        assert np.abs(dw.cpu().numpy() - ref).max() <= tol, splits
    ops.conv2d_wgrad_winograd(xd, Cin, dyd, Cout, dw, S=S, splits=2, scale=0.5, accumulate=True)
    assert np.abs(dw.cpu().numpy() - 1.5 * ref).max() <= 2 * tol


@pytest.mark.parametrize("B", [1, 3, 16, 37])
def test_fc_stream_vs_f64_and_conv_path(hip_lib, B):
    """fc6 as a weight stream (dim_fc_fwd: partial tiles + fixed-order reduce, 32 rows per pass) vs float64 and vs the 8x10
    "convolution" it replaces; MXNet's (c,h,w) flatten order"""
    from lib.hip import ops

    g = torch.Generator().manual_seed(40 + B)
    feat = torch.randn((B, 1024, 8, 10), generator=g)
    w = torch.randn((256, 81920), generator=g) * 0.01
    b = torch.randn((256,), generator=g) * 0.1
    ref = torch.nn.functional.leaky_relu(feat.double().reshape(B, -1) @ w.double().t() + b.double(), 0.1).numpy()
    x = feat.permute(0, 2, 3, 1).contiguous().to("cuda:0")
    wp = ops.fc_pack_weight(w.to("cuda:0"), 1024, 8, 10)
    y = ops.fc_fwd(x, wp, b.to("cuda:0"), 256, slope=0.1)
    np.testing.assert_allclose(y.cpu().numpy(), ref, atol=5e-5, rtol=1e-4)
    y2 = ops.conv2d_fwd(x, wp, b.to("cuda:0"), 256, 8, 10, 1, 0, slope=0.1, splits=40, tile=3).view(B, 256)
    np.testing.assert_allclose(y.cpu().numpy(), y2.cpu().numpy(), atol=5e-5, rtol=1e-4)
    np.testing.assert_array_equal(ops.fc_fwd(x, wp, b.to("cuda:0"), 256, slope=0.1).cpu().numpy(), y.cpu().numpy())  # deterministic


def test_winograd_batch_slices(hip_lib, monkeypatch):
    """batches whose transformed tiles would exceed the 32-bit offsets of the plane GEMMs run as slices of whole images through
    the same workspace; DIM_WINO_MAX_SLICE forces that path at a size the test can check (5 images as 2 + 2 + 1)"""
    from lib.hip import ops

    g = torch.Generator().manual_seed(77)
    x = torch.randn((5, 22, 30, 64), generator=g).to("cuda:0")
    b = (torch.randn((128,), generator=g) * 0.1).to("cuda:0")
    w3 = (torch.randn((128, 64, 3, 3), generator=g) * 0.05).to("cuda:0")
    w5 = (torch.randn((128, 64, 5, 5), generator=g) * 0.03).to("cuda:0")
    wp3, wp5 = ops.winograd_pack_weight(w3, m=4), ops.winograd5x5s2_pack_weight(w5)
    ws = torch.empty(ops.lib().dim_winograd5x5s2_workspace_floats(5, 22, 30, 64, 128) + ops.lib().dim_winograd_workspace_floats(5, 22, 30, 64, 128, 4),
                     device="cuda:0")
    whole3 = ops.conv2d_fwd_winograd(x, 64, wp3, b, 128, m=4, workspace=ws).clone()
    whole5 = ops.conv2d_fwd_winograd5x5s2(x, 64, wp5, b, 128, workspace=ws).clone()
    monkeypatch.setenv("DIM_WINO_MAX_SLICE", "2")
    # (equal up to where the stream-K ranges cut a (tile, plane) item: a slice has other row counts, so other cuts and another GEMM tile
    # -- the two partial sums of a cut item then differ in the last bits; the same call twice is bit-identical)
    s3 = ops.conv2d_fwd_winograd(x, 64, wp3, b, 128, m=4, workspace=ws).clone()
    s5 = ops.conv2d_fwd_winograd5x5s2(x, 64, wp5, b, 128, workspace=ws).clone()
    np.testing.assert_allclose(s3.cpu().numpy(), whole3.cpu().numpy(), atol=2e-5)
    np.testing.assert_allclose(s5.cpu().numpy(), whole5.cpu().numpy(), atol=2e-5)
    np.testing.assert_array_equal(ops.conv2d_fwd_winograd(x, 64, wp3, b, 128, m=4, workspace=ws).cpu().numpy(), s3.cpu().numpy())
    np.testing.assert_array_equal(ops.conv2d_fwd_winograd5x5s2(x, 64, wp5, b, 128, workspace=ws).cpu().numpy(), s5.cpu().numpy())


@pytest.mark.parametrize("shape,tile", [((16, 60, 80, 64, 256, 3, 1, 1), 4), ((9, 120, 160, 32, 128, 3, 2, 1), 4), ((6, 96, 128, 32, 64, 3, 1, 1), 3)])
def test_conv_auto_split_tail_vs_f64(hip_lib, shape, tile):
    """splits = 0: whole tiles per CU in one launch + split-K tail (dim_conv2d_tail_plan confirms the two-launch path) vs float64"""
    import ctypes

    import torch.nn.functional as F
    from lib.hip import ops

    N, H, W, Cin, Cout, k, s, p = shape
    g = torch.Generator().manual_seed(sum(shape))
    x = torch.randn((N, Cin, H, W), generator=g)
    w = torch.randn((Cout, Cin, k, k), generator=g) * (1.0 / np.sqrt(k * k * Cin))
    b = torch.randn((Cout,), generator=g) * 0.1
    ref = F.leaky_relu(F.conv2d(x.double(), w.double(), b.double(), stride=s, padding=p), 0.1).permute(0, 2, 3, 1).numpy()
    Ho, Wo = ref.shape[1:3]
    tb, ts = ctypes.c_int(0), ctypes.c_int(1)
    ops.check(ops.lib().dim_conv2d_tail_plan(N * Ho * Wo, Cout, Cin, k, k, tile, ctypes.byref(tb), ctypes.byref(ts)))
    assert ts.value >= 2 and tb.value > 0, (tb.value, ts.value)
    xd = x.permute(0, 2, 3, 1).contiguous().to("cuda:0")
    wp = ops.conv2d_pack_weight(w.to("cuda:0"))
    ws = torch.full((ops.lib().dim_conv2d_workspace_floats(N, H, W, Cin, Cout, k, k, s, p, 0),), float("nan"), device="cuda:0")
    y = ops.conv2d_fwd(xd, wp, b.to("cuda:0"), Cout, k, k, s, p, slope=0.1, splits=0, tile=tile, workspace=ws).cpu().numpy()
    assert np.isfinite(y).all()
    assert np.abs(y - ref).max() <= 1e-4 * np.abs(ref).max() + 2e-5
    y1 = ops.conv2d_fwd(xd, wp, b.to("cuda:0"), Cout, k, k, s, p, slope=0.1, splits=1, tile=tile).cpu().numpy()
    bm, bn = (128, 128) if tile == 4 else (64, 64)
    head = tb.value // (Cout // bn) * bm                           # rows of the un-split part: bit-identical to the single launch
    np.testing.assert_array_equal(y.reshape(-1, Cout)[:head], y1.reshape(-1, Cout)[:head])


def test_copy_add_rows_and_fill(hip_lib):
    """dim_copy_rows / dim_add_rows / dim_fill_words: channel-range copies and adds between NHWC maps with different channel counts
    (vectorised and scalar paths), bit-exact against torch slicing."""
    from lib.hip import ops

    g = torch.Generator().manual_seed(3)
    for cd, cs, d0, s0, n in ((1088, 512, 0, 0, 512), (832, 512, 0, 0, 512), (7, 4, 0, 0, 4), (7, 3, 4, 0, 3), (20, 9, 5, 2, 6)):
        dst = torch.randn(2, 5, 6, cd, generator=g).to("cuda:0")
        src = torch.randn(2, 5, 6, cs, generator=g).to("cuda:0")
        want = dst.clone()
        want[..., d0:d0 + n] = src[..., s0:s0 + n]
        assert torch.equal(ops.copy_nhwc_channels(dst.clone(), d0, src, s0, n), want)
        want = dst.clone()
        want[..., d0:d0 + n] += src[..., s0:s0 + n]
        assert torch.equal(ops.copy_nhwc_channels(dst.clone(), d0, src, s0, n, add=True), want)
    t = torch.randn(1000, generator=g).to("cuda:0")
    assert torch.equal(ops.fill(t, 0.0), torch.zeros_like(t)) and torch.equal(ops.fill(t, 2.5), torch.full_like(t, 2.5))
    ti = torch.ones(17, dtype=torch.int32, device="cuda:0")
    assert torch.equal(ops.fill(ti, -3), torch.full_like(ti, -3))


@pytest.mark.parametrize("shape", [(2, 60, 80, 64, 128, 0), (1, 30, 40, 256, 512, 0), (2, 15, 20, 128, 64, 3), (1, 29, 37, 32, 64, 0),
                                   (3, 9, 11, 96, 192, 4), (2, 8, 8, 64, 256, 5), (5, 16, 24, 32, 128, 0),
                                   # the few-row GEMM tiles: 16 x (4 x 5) = 320 rows = two 160-row tiles (conv5 at 16 pairs), 7 x 20 = 140
                                   # rows (one partly empty tile), 96-row tiles with 100 and 96 rows
                                   (16, 30, 40, 64, 512, 6), (7, 30, 40, 32, 128, 6), (5, 30, 40, 32, 256, 7), (16, 16, 20, 64, 128, 7)])
def test_conv_winograd3x3s2_vs_f64(hip_lib, shape, monkeypatch, wino_split):
    """3x3 / stride-2 / pad-1 layers through their phase images (minimal filtering: F(4,1) on the even, F(4,2) on the odd phase; 81 plane
    GEMMs) vs torch-CPU float64 and vs the direct kernel: odd and even maps (tiles hanging over both edges), padded channel strides, an
    output channel offset, every GEMM tile, no bias, batch slices"""
    import torch.nn.functional as F
    from lib.hip import ops

    N, H, W, Cin, Cout, tile = shape
    g = torch.Generator().manual_seed(sum(shape) + 5)
    x = torch.randn((N, Cin, H, W), generator=g)
    x = torch.where(x > 0, x, 0.1 * x)                           # post-LeakyReLU statistics, as in the network
    w = torch.randn((Cout, Cin, 3, 3), generator=g) / np.sqrt(9 * Cin)
    b = torch.randn((Cout,), generator=g) * 0.1
    ref = F.leaky_relu(F.conv2d(x.double(), w.double(), b.double(), stride=2, padding=1), 0.1).permute(0, 2, 3, 1).numpy()
    Ho, Wo = ref.shape[1:3]
    xd = torch.zeros((N, H, W, Cin + 8), device="cuda:0")
    xd[..., :Cin] = x.permute(0, 2, 3, 1).to("cuda:0")
    wp = ops.winograd3x3s2_pack_weight(w.to("cuda:0"))
    y = torch.full((N, Ho, Wo, Cout + 16), 7.0, device="cuda:0")
    ops.conv2d_fwd_winograd3x3s2(xd, Cin, wp, b.to("cuda:0"), Cout, slope=0.1, tile=tile, out=y, out_coff=4)
    got = y.cpu().numpy()
    tol = 1e-4 * np.abs(ref).max() + 2e-5
    assert np.abs(got[..., 4:4 + Cout] - ref).max() <= tol
    assert np.all(got[..., :4] == 7.0) and np.all(got[..., 4 + Cout:] == 7.0)
    direct = ops.conv2d_fwd(xd[..., :Cin].contiguous(), ops.conv2d_pack_weight(w.to("cuda:0")), b.to("cuda:0"), Cout, 3, 3, 2, 1, slope=0.1, tile=3)
    assert np.abs(got[..., 4:4 + Cout] - direct.cpu().numpy()).max() <= tol
    ref0 = F.conv2d(x.double(), w.double(), None, stride=2, padding=1).permute(0, 2, 3, 1).numpy()      # no bias, linear
    y0 = ops.conv2d_fwd_winograd3x3s2(xd, Cin, wp, None, Cout, slope=1.0, tile=tile)
    assert np.abs(y0.cpu().numpy() - ref0).max() <= 1e-4 * np.abs(ref0).max() + 2e-5
    if N >= 3:   # the batch in slices of two images through the same workspace
        monkeypatch.setenv("DIM_WINO_MAX_SLICE", "2")
        y2 = ops.conv2d_fwd_winograd3x3s2(xd, Cin, wp, None, Cout, slope=1.0, tile=tile)
        if N * Cout <= 1024:
            assert torch.equal(y2, y0)
        else:   # more (tile, plane) items than resident workgroups: the stream-K ranges cut items at other K chunks in a slice than in
            #     the whole batch, so the two partial sums of a cut item differ in the last bit
            assert (y2 - y0).abs().max().item() <= 2e-6 * y0.abs().max().item()


@pytest.mark.parametrize("shape", [(16, 256, 1024, 8, 10), (1, 256, 1024, 8, 10), (17, 256, 1024, 8, 10), (32, 70, 32, 3, 5), (3, 64, 48, 1, 1)])
def test_fc_wgrad_in_mxnet_layout_vs_f64(hip_lib, shape):
    """fc6's weight gradient written straight in MXNet's (out, c*h*w) layout from the NHWC activation (dim_fc_wgrad_nhwc) vs float64;
    batches on both sides of 16 (two kernel shapes), an output count that is not a multiple of 64, a 1 x 1 map"""
    from lib.hip import ops

    B, Out, C, H, W = shape
    g = torch.Generator().manual_seed(sum(shape))
    x = torch.randn((B, C, H, W), generator=g)
    dz = torch.randn((B, Out), generator=g)
    ref = dz.double().t() @ x.double().reshape(B, -1)
    dW = torch.full((Out, C * H * W), 3.0, device="cuda:0")
    ops.fc_wgrad_nhwc(dz.to("cuda:0"), x.permute(0, 2, 3, 1).contiguous().to("cuda:0"), dW)
    assert (dW.cpu().double() - ref).abs().max().item() <= 1e-5 * ref.abs().max().item() + 1e-6
