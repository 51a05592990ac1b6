"""SURVEY 8(f) N4 data layer, host side (CPU): helpers pinned by vectors generated from the reference's own modules
(tests/golden/make_golden.py: mask_dilate, get_min_rect, backproject_camera, se3_mul / se3_inverse, calc_flow), and the pairdb -> blob
functions of lib/utils/image.py / lib/pair_matching/data_pair.py on a small synthetic dataset written to disk with PIL
(the reference's versions need cv2 and cannot be imported here: parity of those is by restatement, file:line in the docstrings)."""
import os

import numpy as np
import pytest
from PIL import Image

from conftest import ROOT

GOLD = os.path.join(ROOT, "tests", "golden")


def test_helpers_vs_reference_goldens():
    from lib.pair_matching.flow import calc_flow
    from lib.utils.get_min_rect import get_min_rect
    from lib.utils.mask_dilate import mask_dilate
    from lib.utils.projection import backproject_camera, se3_inverse, se3_mul

    g = np.load(os.path.join(GOLD, "data_golden.npz"))
    seen_dirs = set()
    for seed, k, t, out in zip(g["dil_seed"], g["dil_mask"], g["dil_thick"], g["dil_out"]):
        np.random.seed(int(seed))
        seen_dirs.add(np.random.randint(10))
        np.random.seed(int(seed))
        np.testing.assert_array_equal(mask_dilate(g["masks"][k], max_thickness=int(t)), out)
    assert seen_dirs == set(range(10))   # the fixture covers every `direction`
    np.testing.assert_array_equal(backproject_camera(g["depth"], g["K"]), g["backproject"])
    f, v, X = calc_flow(g["cf_depth_src"], g["cf_pose_src"], g["cf_pose_tgt"], g["K"], g["cf_depth_tgt"], standard_rep=True)
    np.testing.assert_allclose(f, g["cf_flow_std"], atol=1e-6)
    np.testing.assert_array_equal(v, g["cf_visible"])
    np.testing.assert_array_equal(X, g["cf_X_valid"])
    fg = np.load(os.path.join(GOLD, "flow_golden.npz"))
    for i in range(len(fg["depth_src"])):
        f, v, _ = calc_flow(fg["depth_src"][i], fg["pose_src"][i], fg["pose_tgt"][i], fg["K"], fg["depth_tgt"][i])
        np.testing.assert_allclose(f, fg["flow"][i], atol=1e-6)     # "[h, w]" order
        np.testing.assert_array_equal(v, fg["visible"][i])
    m = np.load(os.path.join(GOLD, "min_rect_golden.npz"))
    for mask, rect in zip(m["masks"], m["rects"]):
        assert tuple(get_min_rect(mask)) == tuple(rect)
    s = np.load(os.path.join(GOLD, "se3_golden.npz"))
    for a, b, ab, ai in zip(s["pose_src"], s["pose_tgt"], s["se3_mul"], s["se3_inv"]):
        r1, r2 = se3_mul(a, b), se3_inverse(a)
        assert r1.dtype == np.float32 and r2.dtype == np.float32
        np.testing.assert_array_equal(r1, ab)
        np.testing.assert_array_equal(r2, ai)


# ---------------------------------------------------------------------------------------------------------------- synthetic files
H, W = 48, 64


def _write_dataset(root, n=3, seed=0):
    rng = np.random.default_rng(seed)
    os.makedirs(root, exist_ok=True)
    pairdb = []
    for i in range(n):
        y0, x0, h, w = int(rng.integers(4, 14)), int(rng.integers(6, 20)), int(rng.integers(10, 24)), int(rng.integers(12, 30))
        label = np.zeros((H, W), np.uint8)
        label[y0:y0 + h, x0:x0 + w] = 3 + i        # the object's label value = mask_idx
        label[2:5, 50:60] = 9                      # another object in the same label image
        depth_obs = np.where(label == 3 + i, rng.integers(600, 900, size=(H, W)), 0).astype(np.uint16)
        ren = np.zeros((H, W), np.uint16)
        ren[y0 + 2:y0 + h + 1, x0 - 1:x0 + w - 3] = rng.integers(500, 1000, size=(h - 1, w - 2))
        ren[0, 0] = 150                            # a pixel BELOW the 0.2 m threshold: stays 0.15 in mask_rendered, not in the bbox
        obs_rgb = rng.integers(0, 256, size=(H, W, 3)).astype(np.uint8)
        ren_rgb = rng.integers(0, 256, size=(H, W, 3)).astype(np.uint8)
        p = {k: os.path.join(root, "{:02d}-{}.png".format(i, k)) for k in ("color", "color_r", "depth", "depth_r", "label")}
        Image.fromarray(obs_rgb).save(p["color"])
        Image.fromarray(ren_rgb).save(p["color_r"])
        Image.fromarray(depth_obs).save(p["depth"])
        Image.fromarray(ren).save(p["depth_r"])
        Image.fromarray(label).save(p["label"])
        pose_r = np.hstack([np.eye(3), [[0.01 * i], [0.02], [0.7]]])
        pose_o = np.hstack([np.eye(3), [[0.012 * i], [0.018], [0.71]]])
        pairdb.append({"image_observed": p["color"], "image_rendered": p["color_r"], "depth_gt_observed": p["depth"], "depth_observed": p["depth"],
                       "depth_rendered": p["depth_r"], "mask_gt_observed": p["label"], "mask_observed": p["label"], "mask_idx": 3 + i,
                       "pose_observed": pose_o, "pose_rendered": pose_r, "gt_class": ["ape", "can"][i % 2], "height": H, "width": W,
                       "img_flipped": False, "_raw": (obs_rgb, ren_rgb, depth_obs, ren, label)})
    return pairdb


@pytest.fixture()
def cfg():
    from deepim.config.config import config, reset_config

    reset_config()
    config.SCALES = [(H, W)]
    config.network.PIXEL_MEANS = np.array([123.68, 116.779, 103.939])
    config.network.INPUT_MASK = True
    config.network.PRED_MASK = True
    config.network.PRED_FLOW = True
    config.dataset.class_name = ["ape", "can"]
    config.dataset.INTRINSIC_MATRIX = np.array([[57.2, 0, 32.5], [0, 57.3, 24.2], [0, 0, 1]])
    config.TEST.MASK_DILATE = False
    yield config
    reset_config()


def test_readers_and_transform(tmp_path, cfg):
    from lib.utils import image as I

    db = _write_dataset(str(tmp_path))
    obs_rgb, ren_rgb, depth_obs, ren, label = db[0]["_raw"]
    bgr = I.imread_color(db[0]["image_observed"])
    np.testing.assert_array_equal(bgr, obs_rgb[:, :, ::-1])                       # cv2 order: B, G, R
    assert I.imread_unchanged(db[0]["depth_rendered"]).dtype == np.uint16
    np.testing.assert_array_equal(I.imread_unchanged(db[0]["depth_rendered"]), ren)
    np.testing.assert_array_equal(I.imread_unchanged(db[0]["mask_gt_observed"]), label)
    t = I.transform(bgr, cfg.network.PIXEL_MEANS)
    assert t.shape == (1, 3, H, W)
    for c in range(3):  # plane c = channel 2-c of the BGR image minus PIXEL_MEANS[2-c]: plane 0 is RED minus 103.939
        np.testing.assert_allclose(t[0, c], bgr[:, :, 2 - c].astype(np.float64) - cfg.network.PIXEL_MEANS[2 - c])
    np.testing.assert_array_equal(I.transform_inverse(t, cfg.network.PIXEL_MEANS), obs_rgb)
    same, scale = I.resize(bgr, H, W)
    assert scale == 1.0 and same is bgr
    _, scale2 = I.resize(bgr, 2 * H, 2 * W)
    assert scale2 == 2.0
    padded, _ = I.resize(bgr, H, W, stride=32)
    assert padded.shape == (64, 64, 3) and padded[H:].sum() == 0


def test_get_pair_image_depth_mask_test_phase(tmp_path, cfg):
    from lib.utils import image as I

    db = _write_dataset(str(tmp_path))
    obs, ren, scales = I.get_pair_image(db, cfg, "test")
    assert scales == [0, 0, 0] and obs[1].shape == (1, 3, H, W)
    np.testing.assert_allclose(ren[2][0, 0], db[2]["_raw"][1][:, :, 0].astype(np.float64) - cfg.network.PIXEL_MEANS[2])
    # INIT_MASK box_rendered: rectangle [y0:y1, x0:x1] of depth_rendered > 0.2, END-exclusive (one row / column short), and the
    # rendered "mask" is the depth with > 0.2 replaced by 1
    mo, mg, mr = I.get_pair_mask(db, cfg, scales, "test")
    for i, rec in enumerate(db):
        d = rec["_raw"][3].astype(np.float32) / 1000.0
        ys, xs = np.nonzero(d > 0.2)
        box = np.zeros((H, W))
        box[ys.min():ys.max(), xs.min():xs.max()] = 1
        np.testing.assert_array_equal(mo[i][0, 0], box)
        np.testing.assert_array_equal(mg[i][0, 0], box)          # test phase: mask_gt_observed IS mask_observed
        np.testing.assert_allclose(mr[i][0, 0], np.where(d > 0.2, 1.0, d))
        assert mr[i][0, 0, 0, 0] == pytest.approx(0.15)
    for kind, ref in (("mask_gt_observed", lambda r: (r["_raw"][4] == r["mask_idx"]).astype(float)),
                      ("mask_observed", lambda r: (r["_raw"][4] == r["mask_idx"]).astype(float))):
        cfg.TEST.INIT_MASK = kind
        mo, _, _ = I.get_pair_mask(db, cfg, scales, "test")
        np.testing.assert_array_equal(mo[1][0, 0], ref(db[1]))
    for kind in ("box_gt_observed", "box_"):
        cfg.TEST.INIT_MASK = kind
        mo, _, _ = I.get_pair_mask(db, cfg, scales, "test")
        ys, xs = np.nonzero(db[0]["_raw"][4] == db[0]["mask_idx"])
        box = np.zeros((H, W))
        box[ys.min():ys.max(), xs.min():xs.max()] = 1
        np.testing.assert_array_equal(mo[0][0, 0], box)
    cfg.TEST.INIT_MASK = "nonsense"
    with pytest.raises(Exception, match="Unknown init mask type"):
        I.get_pair_mask(db, cfg, scales, "test")
    # an undetected object (all-zero rendered depth) gives an empty observed mask instead of an exception
    cfg.TEST.INIT_MASK = "box_rendered"
    Image.fromarray(np.zeros((H, W), np.uint16)).save(db[0]["depth_rendered"])
    mo, _, mr = I.get_pair_mask(db[:1], cfg, [0], "test")
    assert mo[0].sum() == 0 and mr[0].sum() == 0
    # depth blobs in metres
    d_obs, d_ren = I.get_pair_depth(db[1:2], cfg, [0], "test")
    np.testing.assert_allclose(d_obs[0][0, 0], db[1]["_raw"][2].astype(np.float32) / 1000.0)
    np.testing.assert_allclose(I.get_gt_observed_depth(db[1:2], cfg, [0])[0], d_obs[0])


def test_get_pair_mask_train_phase_and_dilation(tmp_path, cfg):
    from lib.utils import image as I
    from lib.utils.mask_dilate import mask_dilate

    db = _write_dataset(str(tmp_path))
    for kind in ("box_gt", "box_rendered", "mask_gt"):
        cfg.TRAIN.INIT_MASK = kind
        mo, mg, mr = I.get_pair_mask(db, cfg, [0, 0, 0], "train")
        gt = (db[2]["_raw"][4] == db[2]["mask_idx"]).astype(float)
        np.testing.assert_array_equal(mg[2][0, 0], gt)
        if kind == "box_gt":
            ys, xs = np.nonzero(gt)
            box = np.zeros((H, W))
            box[ys.min():ys.max(), xs.min():xs.max()] = 1
            np.testing.assert_array_equal(mo[2][0, 0], box)
        if kind == "mask_gt":   # the RAW label image (values 0 / mask_idx / other labels), as the reference copies it
            np.testing.assert_array_equal(mo[2][0, 0], db[2]["_raw"][4].astype(np.float32))
    cfg.TRAIN.INIT_MASK, cfg.TRAIN.MASK_DILATE = "box_gt", True
    np.random.seed(5)
    mo, _, _ = I.get_pair_mask(db[:1], cfg, [0], "train")
    cfg.TRAIN.MASK_DILATE = False
    plain, _, _ = I.get_pair_mask(db[:1], cfg, [0], "train")
    np.random.seed(5)
    np.testing.assert_array_equal(mo[0][0, 0], mask_dilate(plain[0][0, 0]))
    cfg.TRAIN.INIT_MASK = "other"
    with pytest.raises(Exception, match="Unknown mask type"):
        I.get_pair_mask(db, cfg, [0, 0, 0], "train")


def test_flow_labels_points_and_voc_background(tmp_path, cfg):
    from lib.pair_matching.flow import calc_flow
    from lib.utils import image as I

    db = _write_dataset(str(tmp_path))
    flow, w, Xv, _ = I.get_pair_flow(db, cfg, [0, 0, 0], "train")
    d_r, d_o = db[1]["_raw"][3].astype(np.float32) / 1000, db[1]["_raw"][2].astype(np.float32) / 1000
    f, v, X = calc_flow(d_r, db[1]["pose_rendered"], db[1]["pose_observed"], cfg.dataset.INTRINSIC_MATRIX, d_o)
    assert flow[1].shape == (1, 2, H, W) and w[1].shape == (1, 2, H, W)
    np.testing.assert_array_equal(flow[1][0], f.transpose(2, 0, 1))
    assert (w[1] == 1).all()                                    # FLOW_WEIGHT_TYPE 'all'
    cfg.TRAIN.FLOW_WEIGHT_TYPE = "viz"
    _, w2, _, _ = I.get_pair_flow(db[1:2], cfg, [0], "train")
    np.testing.assert_array_equal(w2[0][0, 0], v)
    np.testing.assert_array_equal(w2[0][0, 1], v)
    # model points: NUM_3D_SAMPLE drawn without replacement, zero-padded with zero weights
    cfg.train_iter.NUM_3D_SAMPLE = 50
    cfg.dataset.model_dir = str(tmp_path)
    os.makedirs(os.path.join(str(tmp_path), "ape"))
    pts = np.random.default_rng(1).normal(size=(30, 3))
    np.savetxt(os.path.join(str(tmp_path), "ape", "points.xyz"), pts)
    I.point_cloud_dict.clear()
    P, Wt = I.get_point_cloud_model(cfg, db)
    assert P[0].shape == (1, 3, 50) and Wt[0][0, :, :30].min() == 1 and Wt[0][0, :, 30:].max() == 0 and np.abs(P[0][0, :, 30:]).max() == 0
    assert sorted(map(tuple, np.round(P[0][0, :, :30].T, 12))) == sorted(map(tuple, np.round(pts, 12)))
    obs = I.get_point_cloud_observed(cfg, P[0][0], db[0]["pose_observed"])
    np.testing.assert_allclose(obs, db[0]["pose_observed"][:, :3] @ P[0][0] + db[0]["pose_observed"][:, 3:4])
    # VOC background behind synthetic observed images (training only): object pixels keep the image, the rest is the background
    voc = os.path.join(str(tmp_path), "VOCdevkit", "VOC2012")
    os.makedirs(os.path.join(voc, "ImageSets", "Main"))
    os.makedirs(os.path.join(voc, "JPEGImages"))
    with open(os.path.join(voc, "ImageSets", "Main", "diningtable_trainval.txt"), "w") as fh:
        fh.write("bg_a  1\nbg_b -1\n")
    bg = np.random.default_rng(2).integers(0, 256, size=(60, 100, 3)).astype(np.uint8)
    Image.fromarray(bg).save(os.path.join(voc, "JPEGImages", "bg_a.jpg"), quality=95)
    cfg.dataset.root_path = str(tmp_path)
    I._voc_lists.clear()
    db[0]["data_syn"] = True
    obs_t, _, _ = I.get_pair_image(db[:1], cfg, "train")
    plain_t, _, _ = I.get_pair_image([{k: v for k, v in db[0].items() if k != "data_syn"}], cfg, "train")
    fg = db[0]["_raw"][4] != 0
    np.testing.assert_array_equal(obs_t[0][0][:, fg], plain_t[0][0][:, fg])
    assert np.abs(obs_t[0][0][:, ~fg] - plain_t[0][0][:, ~fg]).mean() > 10
    canvas = I.fit_background(np.ascontiguousarray(bg[:, :, ::-1]), H, W)
    assert canvas.shape == (H, W, 3) and canvas.dtype == np.uint8 and canvas[:, :W - 2].min() >= 0 and canvas.any()


def test_get_data_pair_test_batch_names_and_shapes(tmp_path, cfg):
    from lib.pair_matching.data_pair import get_data_pair_test_batch

    db = _write_dataset(str(tmp_path))
    data, label, im_info = get_data_pair_test_batch(db, cfg)
    assert label == {} and len(data) == 3 and [tuple(i) for i in im_info] == [(H, W)] * 3
    for i, d in enumerate(data):   # deepim/core/loader.py:35-41 names
        assert set(d) == {"image_observed", "image_rendered", "src_pose", "class_index", "mask_observed", "mask_rendered"}
        assert d["src_pose"].shape == (1, 3, 4) and d["image_observed"].shape == (1, 3, H, W) and d["mask_rendered"].shape == (1, 1, H, W)
        assert int(d["class_index"][0]) == i % 2
