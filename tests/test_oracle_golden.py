"""Pin the CPU oracle against golden vectors produced by the reference's own numpy modules
(tests/golden/make_golden.py).  CPU only."""
import os

import numpy as np
import pytest

from oracle import native, pose_error, se3 as ose3, zoom as ozoom

COORDS = ["MODEL", "CAMERA", "CAMERA_NEW", "NAIVE"]


@pytest.fixture(scope="module")
def g(golden_dir):
    return np.load(os.path.join(golden_dir, "se3_golden.npz"))


def test_quat2mat_docstring_constants(g):
    assert np.allclose(ose3.quat2mat([1, 0, 0, 0]), g["quat2mat_id"])
    assert np.allclose(ose3.quat2mat([0, 1, 0, 0]), g["quat2mat_x180"])
    assert np.allclose(g["quat2mat_x180"], np.diag([1, -1, -1]))
    assert np.allclose(g["euler2quat_123_ryxz"], [0.435953, 0.310622, -0.718287, 0.444435], atol=1e-6)


@pytest.mark.parametrize("coord", COORDS)
def test_RT_transform(g, coord):
    z3, o3 = np.zeros(3), np.ones(3)
    for i in range(len(g["pose_src"])):
        got = ose3.RT_transform(g["pose_src"][i], g["quat_raw"][i], g["trans_delta"][i], z3, o3, coord)
        np.testing.assert_allclose(got, g[coord + "_compose"][i], rtol=0, atol=1e-12 if coord != "NAIVE" else 1e-6)


@pytest.mark.parametrize("coord", COORDS)
def test_calc_RT_delta(g, coord):
    z3, o3 = np.zeros(3), np.ones(3)
    for i in range(len(g["pose_src"])):
        r, t = ose3.calc_RT_delta(g["pose_src"][i], g["pose_tgt"][i], z3, o3, coord, "QUAT")
        np.testing.assert_allclose(r, g[coord + "_delta_q"][i], atol=1e-9 if coord != "NAIVE" else 1e-6)
        np.testing.assert_allclose(t, g[coord + "_delta_t"][i], atol=1e-12 if coord != "NAIVE" else 1e-6)
        assert r[0] >= 0
        rm, _ = ose3.calc_RT_delta(g["pose_src"][i], g["pose_tgt"][i], z3, o3, coord, "MATRIX")
        np.testing.assert_allclose(rm, g[coord + "_delta_R"][i], atol=1e-12 if coord != "NAIVE" else 1e-6)


@pytest.mark.parametrize("coord", ["MODEL", "CAMERA", "CAMERA_NEW", "NAIVE"])
def test_euler_deltas_vs_reference(g, golden_dir, coord):
    """rot_type EULER (RT_transform.py:39-40, :139-140): the oracle's static-xyz restatement against the reference's own outputs"""
    e = np.load(os.path.join(golden_dir, "se3_euler_golden.npz"))
    z3, o3 = np.zeros(3), np.ones(3)
    for i in range(g["pose_src"].shape[0]):
        np.testing.assert_allclose(ose3.euler2mat(*e["euler"][i]), e["e2m_R"][i], atol=1e-14)
        got = ose3.RT_transform(g["pose_src"][i], e["euler"][i], g["trans_delta"][i], z3, o3, coord)
        np.testing.assert_allclose(got, e[coord + "_compose"][i], atol=1e-6 if coord == "NAIVE" else 1e-12)
        r, t = ose3.calc_RT_delta(g["pose_src"][i], g["pose_tgt"][i], z3, o3, coord, "EULER")
        np.testing.assert_allclose(np.array(r), e[coord + "_delta_e"][i], atol=1e-6 if coord == "NAIVE" else 1e-12)
        np.testing.assert_allclose(t, e[coord + "_delta_t"][i], atol=1e-6 if coord == "NAIVE" else 1e-12)
    for R, want in zip(e["lock_R"], e["lock_e"]):   # gimbal lock: third angle := 0
        np.testing.assert_allclose(np.array(ose3.mat2euler(R)), want, atol=1e-12)


def test_means_stds(g):
    m, s = g["T_means2"], g["T_stds2"]
    for i in range(len(g["pose_src"])):
        got = ose3.RT_transform(g["pose_src"][i], g["quat_raw"][i], g["trans_delta"][i], m, s, "CAMERA")
        np.testing.assert_allclose(got, g["ms_CAMERA_compose"][i], atol=1e-12)
        r, t = ose3.calc_RT_delta(g["pose_src"][i], g["pose_tgt"][i], m, s, "CAMERA", "QUAT")
        np.testing.assert_allclose(t, g["ms_CAMERA_delta_t"][i], atol=1e-12)


def test_roundtrip_delta_then_compose(g):
    z3, o3 = np.zeros(3), np.ones(3)
    for coord in ["MODEL", "CAMERA", "CAMERA_NEW"]:
        for i in range(16):
            r, t = ose3.calc_RT_delta(g["pose_src"][i], g["pose_tgt"][i], z3, o3, coord, "QUAT")
            back = ose3.RT_transform(g["pose_src"][i], r, t, z3, o3, coord)
            np.testing.assert_allclose(back, g["pose_tgt"][i], atol=1e-9)


def test_mat2quat_and_dist(g):
    for R, q in zip(g["m2q_R"], g["m2q_q"]):
        np.testing.assert_allclose(ose3.mat2quat(R), q, atol=1e-9)
    for i in range(len(g["pose_src"])):
        rd, td = ose3.calc_rt_dist_m(g["pose_src"][i], g["pose_tgt"][i])
        np.testing.assert_allclose([rd, td], g["rt_dist"][i], rtol=1e-9, atol=1e-9)


def test_se3_mul_inverse(g):
    for i in range(len(g["pose_src"])):
        np.testing.assert_array_equal(ose3.se3_mul(g["pose_src"][i], g["pose_tgt"][i]), g["se3_mul"][i])
        np.testing.assert_array_equal(ose3.se3_inverse(g["pose_src"][i]), g["se3_inv"][i])


def test_pose_error(golden_dir):
    g = np.load(os.path.join(golden_dir, "pose_error_golden.npz"))
    for i in range(len(g["add"])):
        e, gt = g["pose_est"][i], g["pose_gt"][i]
        assert np.isclose(pose_error.add(e[:, :3], e[:, 3], gt[:, :3], gt[:, 3], g["pts"]), g["add"][i], rtol=1e-12)
        assert np.isclose(pose_error.adi(e[:, :3], e[:, 3], gt[:, :3], gt[:, 3], g["pts"]), g["adi"][i], rtol=1e-12)


def test_min_rect_matches_bbox_rule(golden_dir):
    """get_min_rect (reference) == the bbox reduction used by zoom_mask (oracle._bbox)."""
    g = np.load(os.path.join(golden_dir, "min_rect_golden.npz"))
    for m, r in zip(g["masks"], g["rects"]):
        nz_x, nz_y = ozoom._bbox(m > 0)
        assert (nz_x.min(), nz_y.min(), nz_x.max(), nz_y.max()) == tuple(r)


def test_flow_restatement_vs_calc_flow(golden_dir):
    """gpu_flow_kernel.cu restatement vs the reference's numpy calc_flow on pixels where both predicates
    coincide: src depth != 0, projection strictly inside, |dz| not within 1e-5 of the threshold."""
    g = np.load(os.path.join(golden_dir, "flow_golden.npz"))
    K = g["K"]
    Kinv = np.linalg.inv(K).astype(np.float32)
    n = len(g["depth_src"])
    KT = np.zeros((n, 3, 4), dtype=np.float32)
    for i in range(n):
        R, t = ose3.calc_se3(g["pose_src"][i], g["pose_tgt"][i])
        KT[i] = np.dot(K, np.concatenate([R, t.reshape(3, 1)], axis=1)).astype(np.float32)
    flow, valid = native.gpu_flow(g["depth_src"][:, None], g["depth_tgt"][:, None], KT, Kinv)
    assert valid.sum() > 1000
    H, W = g["depth_src"].shape[1:]
    checked = 0
    for i in range(n):
        ref_flow = g["flow"][i]  # (H,W,2) in (dy,dx)
        ref_vis = g["visible"][i]
        f = np.stack([flow[i, 0], flow[i, 1]], axis=-1)
        # predicate differences: kernel needs d_src > 1e-3 and wp in [0, W-1]; calc_flow needs round(wp) in [0, W)
        both = (valid[i, 0] == 1) & (ref_vis == 1)
        np.testing.assert_allclose(f[both], ref_flow[both], atol=2e-3)
        checked += both.sum()
        disagree = (valid[i, 0] != ref_vis).sum()
        assert disagree <= 0.01 * max(1, (ref_vis == 1).sum()), (i, disagree)
    assert checked > 1000
