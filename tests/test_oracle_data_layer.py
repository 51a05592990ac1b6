"""oracle/data_layer.py (the checker of the device data layer) pinned by vectors generated from the reference's own modules
(tests/golden/make_golden.py): mask_dilate under every `direction`, get_min_rect, backproject_camera, calc_flow in both flow
representations.  CPU only."""
import os

import numpy as np

from conftest import ROOT
from oracle import data_layer as odl

GOLD = os.path.join(ROOT, "tests", "golden")


def test_oracle_data_helpers_vs_reference_goldens():
    g = np.load(os.path.join(GOLD, "data_golden.npz"))
    dirs = set()
    for seed, k, t, out in zip(g["dil_seed"], g["dil_mask"], g["dil_thick"], g["dil_out"]):
        np.random.seed(int(seed))
        dirs.add(np.random.randint(10))
        np.random.seed(int(seed))
        np.testing.assert_array_equal(odl.mask_dilate(g["masks"][k], max_thickness=int(t)), out)
    assert dirs == set(range(10))
    np.testing.assert_array_equal(odl.backproject_camera(g["depth"], g["K"]), g["backproject"])
    f, v = odl.calc_flow(g["cf_depth_src"], g["cf_pose_src"], g["cf_pose_tgt"], g["K"], g["cf_depth_tgt"], standard_rep=True)
    np.testing.assert_allclose(f, g["cf_flow_std"], atol=1e-6)
    np.testing.assert_array_equal(v, g["cf_visible"])
    fg = np.load(os.path.join(GOLD, "flow_golden.npz"))
    for i in range(len(fg["depth_src"])):
        f, v = odl.calc_flow(fg["depth_src"][i], fg["pose_src"][i], fg["pose_tgt"][i], fg["K"], fg["depth_tgt"][i])
        np.testing.assert_allclose(f, fg["flow"][i], atol=1e-6)     # "[h, w]" order
        np.testing.assert_array_equal(v, fg["visible"][i])
    m = np.load(os.path.join(GOLD, "min_rect_golden.npz"))
    for mask, rect in zip(m["masks"], m["rects"]):
        assert tuple(odl.min_rect(mask)) == tuple(rect)


def test_oracle_blob_conventions():
    """the unpinnable assembly, checked for internal consistency: channel order / means, mask_rendered = depth with > 0.2 m -> 1, the
    end-exclusive rectangle, the raw-label quirk of TRAIN.INIT_MASK 'mask_gt', flow weights by type"""
    rng = np.random.default_rng(0)
    H, W = 24, 32
    im = rng.integers(0, 256, size=(H, W, 3)).astype(np.uint8)
    pm = np.array([102.9801, 115.9465, 122.7717])
    blob = odl.image_blob(im, pm)
    assert blob.shape == (1, 3, H, W)
    np.testing.assert_allclose(blob[0, 0], im[:, :, 2] - pm[2])     # plane 0 = R of the BGR image minus the R mean
    d = np.zeros((H, W), np.uint16)
    d[5:15, 8:20] = 800
    d[6, 9] = 150                                                   # 0.15 m: below the 0.2 m threshold, keeps its depth value
    label = np.zeros((H, W), np.uint8)
    label[4:12, 10:25] = 3
    m_obs, m_gt, m_ren = odl.masks_train(label, 3, d, 1000.0, "box_rendered", False)
    assert m_ren[0, 0, 7, 10] == 1.0 and abs(m_ren[0, 0, 6, 9] - 0.15) < 1e-6 and m_ren[0, 0, 0, 0] == 0.0
    assert m_obs[0, 0].sum() == (14 - 5) * (19 - 8) and m_obs[0, 0, 5, 8] == 1 and m_obs[0, 0, 14, 8] == 0 and m_obs[0, 0, 5, 19] == 0
    assert m_gt[0, 0].sum() == 8 * 15
    raw_obs, _, _ = odl.masks_train(label, 3, d, 1000.0, "mask_gt", False)
    assert set(np.unique(raw_obs)) == {0.0, 3.0}
    np.random.seed(1)
    dil, _, _ = odl.masks_train(label, 3, d, 1000.0, "mask_gt", True)
    assert set(np.unique(dil)) <= {0.0, 1.0} and dil.sum() >= (label != 0).sum()
    t_obs, t_ren = odl.masks_test(np.zeros((H, W), np.uint16), 1000.0, "box_rendered", False)
    assert t_obs.sum() == 0 and t_ren.sum() == 0                    # undetected object: empty masks, no exception
