"""The reference's plugin-level view: ops invoked by registered name through `Custom(op_type=...)` with string attrs,
plus the renderer / flow / RT_transform faces.  Mirrors the reference's own `__main__` self-checks where it has them
(zoom_trans.py:108-168, transform3d.py:365-493)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import se3 as ose3, zoom as ozoom  # noqa: E402

DEV = "cuda:0"


def cu(a):
    return torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float32).to(DEV)


@pytest.fixture(scope="module")
def Custom(hip_lib):
    assert torch.cuda.is_available()
    from deepim.operator_py import Custom as C

    return C


def test_zoom_trans_selfcheck(Custom):
    """zoom_trans.py:126-168: forward divides (dx,dy) by wx; inverse op round-trips; backward is identity unless b_zoom_grad."""
    rng = np.random.RandomState(1)
    zf = np.repeat(rng.rand(8, 1) + 0.1, 4, axis=1).astype(np.float32)
    td = rng.randn(8, 3).astype(np.float32)
    z = Custom(zoom_factor=cu(zf), trans_delta=cu(td), op_type="ZoomTrans", name="ZoomTrans", b_inv_zoom=False)
    ref = td.copy()
    ref[:, :2] /= zf[:, :1]
    np.testing.assert_allclose(z.cpu().numpy(), ref, rtol=1e-6)
    back = Custom(zoom_factor=cu(zf), trans_delta=z, op_type="ZoomTrans", b_inv_zoom=True)
    np.testing.assert_allclose(back.cpu().numpy(), td, rtol=1e-6, atol=1e-7)
    from deepim.operator_py.zoom_trans import ZoomTransProp

    for inv, zg in ((False, False), (True, True), (False, True)):
        op = ZoomTransProp(str(inv), str(zg)).create_operator(None, None, None)
        g = cu(rng.randn(8, 3))
        in_grad = [torch.empty(8, 4, device=DEV), torch.empty(8, 3, device=DEV)]
        op.backward(["write", "write"], [g], [cu(zf), cu(td)], [z], in_grad, [])
        np.testing.assert_allclose(in_grad[1].cpu().numpy(), ozoom.zoom_trans_backward(zf, g.cpu().numpy(), inv, zg), rtol=1e-6)
        assert in_grad[0].abs().sum() == 0


def test_zoom_mask_and_image_by_name(Custom):
    from lib.utils.synthetic import LINEMOD_K, PIXEL_MEANS
    from test_gpu_ops import _rand_masks

    rng = np.random.RandomState(11)
    B, H, W = 2, 480, 640
    mo, mr, pose = _rand_masks(rng, B, H, W)
    outs = Custom(mask_observed=cu(mo), mask_gt_observed=cu(mo), mask_rendered=cu(mr), src_pose=cu(pose), K=LINEMOD_K.flatten(),
                  name="ZoomMask", op_type="ZoomMask", height=480, width=640)
    zmo, zmg, zmr, zf = ozoom.zoom_mask(mo, mo, mr, pose, LINEMOD_K, H, W)
    np.testing.assert_allclose(outs[3].cpu().numpy(), zf, rtol=2e-6, atol=2e-6)
    for got, want in zip(outs[:3], (zmo, zmg, zmr)):
        assert (got.cpu().numpy() != want).sum() <= 64  # factor last-bit differences can flip a few border pixels
    io = (rng.randint(0, 256, size=(B, 3, H, W)) - PIXEL_MEANS[::-1].reshape(1, 3, 1, 1)).astype(np.float32)
    zi = Custom(zoom_factor=cu(zf), image_observed=cu(io), image_rendered=cu(io), op_type="ZoomImageWithFactor", height=480, width=640,
                pixel_means=PIXEL_MEANS.flatten())
    ref, _ = ozoom.zoom_image_with_factor(zf, io, io, PIXEL_MEANS, H, W)
    np.testing.assert_allclose(zi[0].cpu().numpy(), ref, atol=2e-3)
    np.testing.assert_array_equal(zi[0].cpu().numpy(), zi[1].cpu().numpy())
    inv = Custom(zoom_factor=cu(zf), mask=cu(zmr), op_type="ZoomMaskWithFactor", height=480, width=640, b_inv_zoom=True)
    assert (inv.cpu().numpy() != ozoom.zoom_mask_with_factor(zf, zmr, True)).sum() <= 8


def test_zoom_mask_empty_observed_raises_like_reference(Custom):
    from lib.utils.synthetic import LINEMOD_K

    z = torch.zeros((1, 1, 480, 640), device=DEV)
    pose = cu(np.array([[[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 1, 1.0]]]))
    with pytest.raises(ValueError):  # np.min of an empty array in the reference (zoom_mask.py:58)
        Custom(mask_observed=z, mask_gt_observed=z, mask_rendered=z, src_pose=pose, K=LINEMOD_K.flatten(), op_type="ZoomMask",
               height=480, width=640)


def test_transform3d_selfcheck(Custom):
    """transform3d.py:365-493: forward vs RT_transform + matmul, |d| < 1e-4."""
    from test_oracle_transform3d import _inputs

    pts, rot, trans, pose_src = _inputs()
    z3, o3 = np.zeros(3, np.float32), np.ones(3, np.float32)
    out = Custom(point_cloud=cu(pts), rotation=cu(rot), translation=cu(trans), pose_src=cu(pose_src), name="Transform3D",
                 op_type="Transform3D", T_means=z3, T_stds=o3, rot_coord="CAMERA")
    for b in range(pts.shape[0]):
        pose = ose3.RT_transform(pose_src[b], rot[b], trans[b], z3, o3, "CAMERA")
        assert np.abs(out[b].cpu().numpy() - (pose[:, :3] @ pts[b] + pose[:, 3:4])).max() < 1e-4


def test_gpu_flow_reference_face(hip_lib, golden_dir):
    from lib.flow_c.flow import gpu_flow, gpu_flow_wrapper
    from oracle import native

    g = np.load(os.path.join(golden_dir, "flow_golden.npz"))
    K = g["K"]
    Kinv = np.linalg.inv(K).astype(np.float32)
    n = len(g["depth_src"])
    KT = np.zeros((n, 3, 4), dtype=np.float32)
    for i in range(n):
        R, t = ose3.calc_se3(g["pose_src"][i], g["pose_tgt"][i])
        KT[i] = np.dot(K, np.concatenate([R, t.reshape(3, 1)], axis=1)).astype(np.float32)
    ds, dt = np.ascontiguousarray(g["depth_src"][:, None]), np.ascontiguousarray(g["depth_tgt"][:, None])
    flow, valid = gpu_flow_wrapper(0)(ds, dt, KT, Kinv)
    rflow, rvalid = native.gpu_flow(ds, dt, KT, Kinv)
    assert flow.shape == (n, 2) + ds.shape[2:] and valid.shape == ds.shape and flow.dtype == np.float32
    assert (valid != rvalid).sum() <= 20
    with pytest.raises(ValueError):
        gpu_flow(ds.astype(np.float64), dt, KT, Kinv)  # the Cython binding rejects non-float32 buffers
    f0, v0 = gpu_flow(ds[:0], dt[:0], KT[:0], Kinv)
    assert f0.shape[0] == 0 and v0.shape[0] == 0


def test_RT_transform_reference_face(hip_lib, golden_dir):
    from lib.pair_matching.RT_transform import RT_transform, calc_RT_delta

    g = np.load(os.path.join(golden_dir, "se3_golden.npz"))
    z3, o3 = np.zeros(3), np.ones(3)
    for i in range(4):
        got = RT_transform(g["pose_src"][i], g["quat_raw"][i], g["trans_delta"][i], z3, o3, "CAMERA")
        np.testing.assert_allclose(got, g["CAMERA_compose"][i], atol=2e-6)
        q, t = calc_RT_delta(g["pose_src"][i], g["pose_tgt"][i], z3, o3, "CAMERA", "QUAT")
        np.testing.assert_allclose(q, g["CAMERA_delta_q"][i], atol=2e-6)
        np.testing.assert_allclose(t, g["CAMERA_delta_t"][i], atol=2e-6)
