"""SURVEY 8(f) N4 on the GPU box: the device half of the data layer (deepim/core/loader.py + csrc/data.hip) against the host form
(lib/pair_matching/data_pair.get_data_pair_test_batch), the double-buffered hand-over into the resident refinement loop, and the
training batch assembly (labels through the device SE(3) kernels)."""
import os

import numpy as np
import pytest
import torch
from PIL import Image

pytestmark = pytest.mark.gpu

from oracle import se3 as ose3  # noqa: E402
from scene import make_test_config, make_train_config  # noqa: E402

DEV = "cuda:0"
H, W = 480, 640


def _write_pairs(root, n, seed=0):
    rng = np.random.default_rng(seed)
    os.makedirs(root, exist_ok=True)
    db = []
    for i in range(n):
        y0, x0, h, w = int(rng.integers(40, 200)), int(rng.integers(60, 300)), int(rng.integers(60, 200)), int(rng.integers(60, 250))
        ren = np.zeros((H, W), np.uint16)
        ren[y0:y0 + h, x0:x0 + w] = rng.integers(500, 1200, size=(h, w))
        ren[3, 5] = 120   # below the 0.2 m threshold
        label = np.zeros((H, W), np.uint8)
        label[y0 + 3:y0 + h - 2, x0 + 4:x0 + w + 5] = 1
        dep = np.where(label == 1, rng.integers(600, 900, size=(H, W)), 0).astype(np.uint16)
        p = {k: os.path.join(root, "{:03d}-{}.png".format(i, k)) for k in ("color", "color_r", "depth", "depth_r", "label")}
        raw_o, raw_r = rng.integers(0, 256, size=(H, W, 3)).astype(np.uint8), rng.integers(0, 256, size=(H, W, 3)).astype(np.uint8)
        Image.fromarray(raw_o).save(p["color"], compress_level=1)
        Image.fromarray(raw_r).save(p["color_r"], compress_level=1)
        Image.fromarray(dep).save(p["depth"])
        Image.fromarray(ren).save(p["depth_r"])
        Image.fromarray(label).save(p["label"])
        pose_r = np.hstack([np.eye(3), [[0.01 * i], [0.02], [0.7 + 0.01 * i]]]).astype(np.float32)
        pose_o = np.hstack([np.eye(3), [[0.012 * i], [0.018], [0.71]]]).astype(np.float32)
        db.append({"image_observed": p["color"], "image_rendered": p["color_r"], "depth_gt_observed": p["depth"], "depth_rendered": p["depth_r"],
                   "mask_gt_observed": p["label"], "mask_idx": 1, "pose_observed": pose_o, "pose_rendered": pose_r,
                   "gt_class": ["ape", "can", "cat"][i % 3], "height": H, "width": W, "img_flipped": False})
    return db


def test_device_loader_matches_host_batches(hip_lib, tmp_path):
    from deepim.core.loader import TestDataLoader
    from lib.pair_matching.data_pair import get_data_pair_test_batch

    cfg = make_test_config(test_iter=1)
    cfg.dataset.class_name = ["ape", "can", "cat"]
    cfg.TEST.MASK_DILATE = False
    db = _write_pairs(str(tmp_path), 7)
    B = 2
    loader = TestDataLoader(db, cfg, batch_size=B, device=DEV, workers=4)
    assert len(loader) == 3 and loader.data_name == ["image_observed", "image_rendered", "src_pose", "class_index", "mask_observed", "mask_rendered"]
    seen = 0
    for k, batch in enumerate(loader):
        host, _, _ = get_data_pair_test_batch(db[k * B:(k + 1) * B], cfg)
        for j in range(B):
            for name in ("image_observed", "image_rendered"):
                np.testing.assert_allclose(batch[name][j].cpu().numpy(), host[j][name][0], atol=1e-5)   # f32(u8) - f32(mean) vs f64
            np.testing.assert_array_equal(batch["mask_observed"][j].cpu().numpy(), host[j]["mask_observed"][0].astype(np.float32))
            np.testing.assert_array_equal(batch["mask_rendered"][j].cpu().numpy(), host[j]["mask_rendered"][0].astype(np.float32))
            np.testing.assert_array_equal(batch["src_pose"][j].cpu().numpy(), host[j]["src_pose"][0].astype(np.float32))
            assert int(batch["class_index"][j]) == int(host[j]["class_index"][0])
            np.testing.assert_array_equal(batch["pose_observed"][j].cpu().numpy(), db[k * B + j]["pose_observed"])
        assert batch["mask_rendered"][0, 0, 3, 5].item() == pytest.approx(0.12)   # the depth itself below the threshold
        seen += 1
    assert seen == 3        # 7 pairs: three whole batches, the remainder is not padded
    loader.reset()
    assert loader.iter_next()
    loader.close()
    cfg.TEST.INIT_MASK = "no_such_kind"
    with pytest.raises(Exception, match="Unknown init mask type"):
        TestDataLoader(db, cfg, batch_size=B, device=DEV)
    cfg.TEST.INIT_MASK = "box_rendered"
    cfg.dataset.class_name = ["ape"]


def test_staged_batches_feed_the_resident_loop(hip_lib):
    """three different synthetic batches go host RAM -> pinned staging -> copy stream -> dim_test_blobs_from_raw -> Refiner (graph replay)
    and give the poses of the direct `load` of the same pixels; batches arrive in order although two staging sets alternate."""
    from deepim.core.loader import ArraySource, TestDataLoader, raw_from_device_batch
    from deepim.core.tester import Predictor, Refiner
    from deepim.symbols.deepIM_flownet import deepIM_flownet
    from lib.render_hip.render_py_multi import Render_Py
    from lib.utils import synthetic as syn

    cfg = make_test_config(test_iter=2)
    cfg.TEST.MASK_DILATE = False
    sym = deepIM_flownet()
    sym.get_symbol(cfg, is_train=False)
    params = sym.init_weights(cfg, {}, {}, seed=0)
    from loop_parity import moving_head

    moving_head(params, seed=1)   # 3-12 deg per iteration: the staged blobs must be the direct ones to the bit, or iteration 2 shows it
    B = 2
    models = syn.make_models(seed=2333, n_models=1, subdiv=3)
    rm = Render_Py(None, cfg.dataset.class_name, syn.LINEMOD_K, meshes=models)
    raws, direct = [], []
    pred = Predictor(cfg, params, B)
    ref = Refiner(cfg, pred, rm, B, capture_graph=True)
    for k in range(3):
        b = syn.build_device_batch(rm, B, seed=50 + k)
        depth = torch.empty((B, 1, H, W), device=DEV)
        rm.render_batch(b["class_index"], b["src_pose"], depth=depth)
        raws.append(raw_from_device_batch(b, cfg.network.PIXEL_MEANS, depth))
        # what the files would hold is 1 mm-quantised depth: build the expected masks from the same quantised values
        dq = torch.from_numpy(raws[-1][2].astype(np.float32) / 1000.0).to(DEV)[:, None]
        mr = torch.where(dq > 0.2, torch.ones_like(dq), dq)
        mo = torch.zeros_like(mr)
        for j in range(B):
            ys, xs = torch.nonzero(dq[j, 0] > 0.2, as_tuple=True)
            mo[j, 0, ys.min():ys.max(), xs.min():xs.max()] = 1
        ref.load(b["image_observed"], b["image_rendered"], mo, mr, b["src_pose"], b["class_index"])
        direct.append(ref.refine().cpu().numpy().copy())
    cat = [np.concatenate([r[i] for r in raws]) for i in range(6)]
    loader = TestDataLoader(None, cfg, batch_size=B, device=DEV, workers=2, source=ArraySource(*cat))
    for k in range(3):
        st = loader.next_raw()
        ref.load_staged(loader, st)
        np.testing.assert_array_equal(ref.refine().cpu().numpy(), direct[k])
    assert not loader.iter_next()
    loader.close()


def test_train_batch_assembly_labels(hip_lib, tmp_path):
    """get_data_pair_train_batch: blob names / shapes of deepim/core/loader.py:164-193, SE(3) labels (device kernels) against the oracle's
    restatement of calc_RT_delta, flow labels from calc_flow, observed points = pose_observed applied to the model sample"""
    from lib.pair_matching.data_pair import get_data_pair_train_batch
    from lib.pair_matching.flow import calc_flow
    from lib.utils import image as I

    cfg = make_train_config()
    cfg.dataset.class_name = ["ape", "can", "cat"]
    cfg.TRAIN.INIT_MASK = "box_gt"
    cfg.TRAIN.MASK_DILATE = True
    cfg.train_iter.NUM_3D_SAMPLE = 100
    cfg.dataset.model_dir = str(tmp_path)
    os.makedirs(os.path.join(str(tmp_path), "ape"))
    np.savetxt(os.path.join(str(tmp_path), "ape", "points.xyz"), np.random.default_rng(1).normal(size=(300, 3)) * 0.05)
    I.point_cloud_dict.clear()
    db = _write_pairs(os.path.join(str(tmp_path), "imgs"), 2)
    np.random.seed(3)
    out = get_data_pair_train_batch(db, cfg)
    data, label = out["data"], out["label"]
    assert set(data) == {"image_observed", "image_rendered", "depth_gt_observed", "class_index", "src_pose", "tgt_pose", "mask_observed",
                         "mask_rendered"}
    assert set(label) == {"rot", "trans", "mask_gt_observed", "flow", "flow_weights", "point_cloud_model", "point_cloud_weights",
                          "point_cloud_observed"}
    assert data["image_observed"].shape == (2, 3, H, W) and label["flow"].shape == (2, 2, H, W) and label["point_cloud_model"].shape == (2, 3, 100)
    assert data["class_index"].tolist() == [0, 1]
    z3, o3 = np.zeros(3), np.ones(3)
    for i, rec in enumerate(db):
        r, t = ose3.calc_RT_delta(rec["pose_rendered"].astype(np.float64), rec["pose_observed"].astype(np.float64), z3, o3, "CAMERA", "QUAT")
        np.testing.assert_allclose(label["rot"][i], r, atol=2e-6)
        np.testing.assert_allclose(label["trans"][i], t, atol=2e-6)
        d_r = I.imread_unchanged(rec["depth_rendered"]).astype(np.float32) / 1000
        d_o = I.imread_unchanged(rec["depth_gt_observed"]).astype(np.float32) / 1000
        f, v, _ = calc_flow(d_r, rec["pose_rendered"], rec["pose_observed"], cfg.dataset.INTRINSIC_MATRIX, d_o)
        np.testing.assert_array_equal(label["flow"][i], f.transpose(2, 0, 1))
        np.testing.assert_allclose(label["point_cloud_observed"][i],
                                   rec["pose_observed"][:, :3] @ label["point_cloud_model"][i] + rec["pose_observed"][:, 3:4], atol=1e-6)
    cfg.dataset.class_name = ["ape"]


# ---------------------------------------------------------------------------------------------------------------- oracle-checked loaders
def _decoded(rec, extra=()):
    """what cv2.imread would hand the reference's getters (decoded with PIL here; the decoders are not under test)"""
    from lib.utils.image import imread_color, imread_unchanged

    raw = {"image_observed": imread_color(rec["image_observed"]), "image_rendered": imread_color(rec["image_rendered"]),
           "depth_rendered": imread_unchanged(rec["depth_rendered"]), "mask_idx": rec.get("mask_idx", 1),
           "pose_rendered": rec["pose_rendered"], "pose_observed": rec["pose_observed"]}
    for k in ("depth_gt_observed", "mask_gt_observed", "depth_observed", "mask_observed") + tuple(extra):
        if k in rec:
            raw[k] = imread_unchanged(rec[k])
    return raw


def _oracle_cfg(cfg, phase):
    t = cfg.TRAIN if phase == "train" else cfg.TEST
    return {"pixel_means": np.asarray(cfg.network.PIXEL_MEANS, np.float64).reshape(3), "depth_factor": cfg.dataset.DEPTH_FACTOR,
            "K": np.asarray(cfg.dataset.INTRINSIC_MATRIX), "init_mask": t.INIT_MASK, "mask_dilate": bool(t.get("MASK_DILATE", False)),
            "input_depth": cfg.network.INPUT_DEPTH, "input_mask": cfg.network.INPUT_MASK, "pred_mask": cfg.network.PRED_MASK,
            "pred_flow": cfg.network.PRED_FLOW, "pm_loss": cfg.train_iter.SE3_PM_LOSS, "num_3d_sample": int(cfg.train_iter.NUM_3D_SAMPLE),
            "flow_weight_type": cfg.TRAIN.FLOW_WEIGHT_TYPE, "standard_flow_rep": cfg.network.STANDARD_FLOW_REP,
            "rot_coord": cfg.network.ROT_COORD, "rot_type": cfg.network.ROT_TYPE, "trans_means": np.asarray(cfg.dataset.trans_means, np.float64),
            "trans_stds": np.asarray(cfg.dataset.trans_stds, np.float64)}


@pytest.mark.parametrize("kind,dilate,input_depth", [("mask_gt_observed", False, False), ("box_gt_observed", True, False),
                                                     ("mask_observed", True, True), ("box_", False, True), ("box_rendered", True, True)])
def test_test_loader_variants_vs_oracle(hip_lib, tmp_path, kind, dilate, input_depth):
    """every TEST.INIT_MASK kind of get_pair_mask (image.py:367-476), TEST.MASK_DILATE and INPUT_DEPTH (get_pair_depth :210-269), built on
    the device from the file pixels, against oracle/data_layer.py (pinned by the reference's own outputs: tests/test_oracle_data_layer.py).
    Pair 2's rendered depth is all zero = an undetected object: empty mask whatever the kind."""
    from deepim.core.loader import TestDataLoader
    from oracle import data_layer as odl

    cfg = make_test_config(test_iter=1)
    cfg.dataset.class_name = ["ape", "can", "cat"]
    cfg.TEST.INIT_MASK, cfg.TEST.MASK_DILATE, cfg.network.INPUT_DEPTH = kind, dilate, input_depth
    db = _write_pairs(str(tmp_path), 4)
    for i, rec in enumerate(db):
        rec["depth_observed"] = rec["depth_gt_observed"]
        rec["mask_observed"] = rec["mask_gt_observed"]
    Image.fromarray(np.zeros((H, W), np.uint16)).save(db[2]["depth_rendered"])
    B = 2
    np.random.seed(11)
    loader = TestDataLoader(db, cfg, batch_size=B, device=DEV, workers=4)
    assert ("depth_observed" in loader.data_name) == input_depth
    got = [{k: v.cpu().numpy().copy() for k, v in batch.items()} for batch in loader]
    loader.close()
    np.random.seed(11)
    ocfg = _oracle_cfg(cfg, "test")
    for i, rec in enumerate(db):
        raw = _decoded(rec)
        raw["class_index"] = cfg.dataset.class_name.index(rec["gt_class"])
        want = odl.test_pair(raw, ocfg)
        b, j = got[i // B], i % B
        for name in ("image_observed", "image_rendered"):
            np.testing.assert_allclose(b[name][j], want[name][0], atol=1e-5)
        for name in ("mask_observed", "mask_rendered") + (("depth_observed", "depth_rendered") if input_depth else ()):
            np.testing.assert_array_equal(b[name][j], want[name][0].astype(np.float32), err_msg="{} pair {}".format(name, i))
        if i == 2:
            assert b["mask_observed"][j].sum() == 0
        elif dilate:
            assert b["mask_observed"][j].sum() > 0
    cfg.TEST.INIT_MASK, cfg.TEST.MASK_DILATE, cfg.network.INPUT_DEPTH = "box_rendered", False, False
    cfg.dataset.class_name = ["ape"]


def _voc_tree(root, n=3, seed=5):
    rng = np.random.default_rng(seed)
    voc = os.path.join(root, "VOCdevkit", "VOC2012")
    os.makedirs(os.path.join(voc, "ImageSets", "Main"))
    os.makedirs(os.path.join(voc, "JPEGImages"))
    rows = []
    for i in range(n):
        name = "2008_{:06d}".format(i)
        Image.fromarray(rng.integers(0, 256, size=(375 + 40 * i, 500, 3)).astype(np.uint8)).save(os.path.join(voc, "JPEGImages", name + ".jpg"), quality=95)
        rows.append("{}  1".format(name))
    rows.append("2008_999999 -1")
    with open(os.path.join(voc, "ImageSets", "Main", "diningtable_trainval.txt"), "w") as f:
        f.write("\n".join(rows) + "\n")


@pytest.mark.parametrize("init_mask,dilate,weights,input_depth,rot_type", [("box_rendered", False, "viz", False, "QUAT"),
                                                                          ("box_gt", True, "valid", True, "QUAT"),
                                                                          ("mask_gt", True, "all", False, "MATRIX")])
def test_train_loader_vs_oracle_and_pixel_cache(hip_lib, tmp_path, monkeypatch, init_mask, dilate, weights, input_depth, rot_type):
    """TrainDataLoader (reference deepim/core/loader.py:120-421 -> data_pair.py:144-265): every data and label blob of a training batch
    built in HBM from the file pixels, against oracle/data_layer.train_pair with the same draws in the same order; pairs 1 and 3 are
    synthetic (`data_syn`: VOC background pasted behind the object); class 'can' has fewer model points than NUM_3D_SAMPLE (zero-padded
    sample).  Second epoch: every file comes from the HBM pixel cache -- no decode -- and gives the same blobs."""
    import random

    from deepim.core import loader as L
    from lib.utils import image as I
    from oracle import data_layer as odl

    cfg = make_train_config()
    cfg.dataset.class_name = ["ape", "can", "cat"]
    cfg.TRAIN.INIT_MASK, cfg.TRAIN.MASK_DILATE, cfg.TRAIN.FLOW_WEIGHT_TYPE = init_mask, dilate, weights
    cfg.network.INPUT_DEPTH, cfg.network.ROT_TYPE = input_depth, rot_type
    cfg.train_iter.NUM_3D_SAMPLE = 1000
    cfg.TRAIN.REPLACE_OBSERVED_BG_RATIO = 0.5
    root = str(tmp_path)
    cfg.dataset.model_dir, cfg.dataset.root_path = os.path.join(root, "models"), root
    rng = np.random.default_rng(1)
    pts = {}
    for cls, n in (("ape", 1500), ("can", 700), ("cat", 1000)):
        os.makedirs(os.path.join(cfg.dataset.model_dir, cls))
        pts[cls] = rng.normal(size=(n, 3)) * 0.05
        np.savetxt(os.path.join(cfg.dataset.model_dir, cls, "points.xyz"), pts[cls])
        pts[cls] = np.loadtxt(os.path.join(cfg.dataset.model_dir, cls, "points.xyz"))
    I.point_cloud_dict.clear()
    I._voc_lists.clear()
    _voc_tree(root)
    db = _write_pairs(os.path.join(root, "imgs"), 4)
    for i, rec in enumerate(db):
        # depths consistent with the two poses (identity rotations, pose_r z = 0.70 + 0.01 i, pose_o z = 0.71): a fronto-parallel patch,
        # so that calc_flow finds most rendered pixels visible in the observed depth
        ren = np.asarray(Image.open(rec["depth_rendered"])).astype(np.uint16)
        Image.fromarray(np.where(ren > 200, 700 + 10 * i, ren).astype(np.uint16)).save(rec["depth_rendered"])
        lab = np.asarray(Image.open(rec["mask_gt_observed"]))
        Image.fromarray(np.where(lab == 1, 710, 0).astype(np.uint16)).save(rec["depth_gt_observed"])
        rec["depth_observed"] = rec["depth_gt_observed"]
        if i in (1, 3):
            rec["data_syn"] = True
        elif i == 2:
            rec["data_syn"] = False   # real image: background replaced with probability REPLACE_OBSERVED_BG_RATIO
    B = 2
    decodes = {"n": 0}
    real_c, real_u = L._imread_color, L._imread_unchanged
    monkeypatch.setattr(L, "_imread_color", lambda p: (decodes.__setitem__("n", decodes["n"] + 1), real_c(p))[1])
    monkeypatch.setattr(L, "_imread_unchanged", lambda p: (decodes.__setitem__("n", decodes["n"] + 1), real_u(p))[1])
    cache = L.PixelCache(DEV, budget_bytes=1 << 30)
    loader = L.TrainDataLoader(None, db, cfg, batch_size=B, shuffle=False, device=DEV, workers=4, cache=cache)
    assert loader.data_name[:6] == ["image_observed", "image_rendered", "depth_gt_observed", "class_index", "src_pose", "tgt_pose"]
    assert loader.label_name == ["rot", "trans", "mask_gt_observed", "flow", "flow_weights", "point_cloud_model", "point_cloud_weights",
                                 "point_cloud_observed"]
    epoch1 = [{k: v.cpu().numpy().copy() for k, v in batch.items()} for batch in loader]
    n_dec1 = decodes["n"]
    assert len(epoch1) == 2 and n_dec1 > 0
    # ---- the oracle with the same draws in the same order (the loader seeds like the reference: loader.py:203-208)
    random.seed(6)
    np.random.seed(3)
    rseed = np.random.randint(999999, size=[99999])
    np.random.seed(rseed[0])
    ocfg = _oracle_cfg(cfg, "train")
    n_bg = 0
    for i, rec in enumerate(db):
        np.random.randint(18)
        random.randrange(len(cfg.SCALES))
        bg = None
        if "data_syn" in rec and (rec["data_syn"] is True or np.random.rand() < cfg.TRAIN.REPLACE_OBSERVED_BG_RATIO):
            voc_root, names = I._voc_backgrounds(cfg)
            pick = names[random.randint(0, len(names) - 1)]
            bg = I.fit_background(I.imread_color(os.path.join(voc_root, "JPEGImages/{}.jpg".format(pick))), H, W)
            n_bg += 1
        raw = _decoded(rec)
        raw["class_index"] = cfg.dataset.class_name.index(rec["gt_class"])
        data, label = odl.train_pair(raw, ocfg, points_obj=pts[rec["gt_class"]], bg_fitted=bg)
        b, j = epoch1[i // B], i % B
        for name in ("image_observed", "image_rendered"):
            np.testing.assert_allclose(b[name][j], data[name][0], atol=1e-5, err_msg="{} pair {}".format(name, i))
        exact = ["depth_gt_observed", "mask_observed", "mask_rendered", "mask_gt_observed", "point_cloud_weights"] + \
            (["depth_observed", "depth_rendered"] if input_depth else [])
        for name in exact:
            want = (data[name] if name in data else label[name])[0].astype(np.float32)
            np.testing.assert_array_equal(b[name][j], want, err_msg="{} pair {}".format(name, i))
        np.testing.assert_array_equal(b["src_pose"][j], data["src_pose"][0].astype(np.float32))
        np.testing.assert_array_equal(b["tgt_pose"][j], data["tgt_pose"][0].astype(np.float32))
        assert int(b["class_index"][j]) == int(data["class_index"][0])
        np.testing.assert_allclose(b["rot"][j], label["rot"][0], atol=2e-6)
        np.testing.assert_allclose(b["trans"][j], label["trans"][0], atol=2e-6)
        # float64 per pixel on both sides; the blob is float32
        np.testing.assert_allclose(b["flow"][j], label["flow"][0], atol=2e-5, err_msg="flow pair {}".format(i))
        assert (b["flow_weights"][j] != label["flow_weights"][0]).sum() == 0
        assert np.abs(label["flow"][0]).max() > 0.5 and 0 < label["flow_weights"][0].mean() <= 1
        np.testing.assert_allclose(b["point_cloud_model"][j], label["point_cloud_model"][0], atol=1e-7)
        np.testing.assert_allclose(b["point_cloud_observed"][j], label["point_cloud_observed"][0], atol=1e-6)
        if rec["gt_class"] == "can":
            assert b["point_cloud_weights"][j][:, 700:].sum() == 0 and b["point_cloud_weights"][j][:, :700].min() == 1
    assert n_bg >= 2
    # ---- second epoch: served from the HBM pixel cache (the random VOC pick may name a background not seen before)
    hits0 = cache.hits
    loader.reset()
    epoch2 = [{k: v.cpu().numpy().copy() for k, v in batch.items()} for batch in loader]
    loader.close()
    per_pair = 5 + int(input_depth)
    assert cache.hits - hits0 >= per_pair * len(db) and decodes["n"] - n_dec1 <= 3, (cache.hits - hits0, decodes["n"] - n_dec1)
    for b1, b2 in zip(epoch1, epoch2):
        for name in ("image_rendered", "depth_gt_observed", "mask_rendered", "mask_gt_observed", "flow", "flow_weights", "rot", "trans"):
            np.testing.assert_array_equal(b1[name], b2[name])
    cfg.network.INPUT_DEPTH, cfg.network.ROT_TYPE = False, "QUAT"
    cfg.dataset.class_name = ["ape"]


def test_training_from_files_equals_training_from_resident_copies(hip_lib, tmp_path):
    """fit_batch fed straight from TrainDataLoader (staging sets re-used while the previous batch is still being trained on, pixel
    cache from the second epoch on) ends in the same weights, bit for bit, as fit_batch fed private copies of the same batches --
    i.e. nothing the loader hands out is overwritten while the training step still reads it.  Dataset: the synthetic LINEMOD-shaped
    tree of lib/dataset/synthetic_files.py (PNG files + points.xyz)."""
    from deepim.core.loader import PixelCache, TrainDataLoader
    from deepim.core.module import MutableModule, fit_batch
    from deepim.symbols.deepIM_flownet import deepIM_flownet
    from lib.dataset.synthetic_files import write_synthetic_dataset
    from lib.pair_matching.batch_updater_py_multi import batchUpdaterPyMulti
    from lib.render_hip.render_py_multi import Render_Py
    from lib.utils import image as I
    from lib.utils import synthetic as syn

    cfg = make_train_config()
    cfg.dataset.class_name = ["ape", "can"]
    cfg.TRAIN.INIT_MASK, cfg.TRAIN.MASK_DILATE, cfg.TRAIN.FLOW_WEIGHT_TYPE = "box_rendered", True, "viz"
    cfg.network.TRAIN_ITER_SIZE = 2
    cfg.train_iter.NUM_3D_SAMPLE = 500
    models = syn.make_models(seed=2333, n_models=2, subdiv=3)
    rm = Render_Py(None, cfg.dataset.class_name, syn.LINEMOD_K, meshes=models)
    B, n = 2, 6
    db = write_synthetic_dataset(str(tmp_path), rm, models, cfg.dataset.class_name, n, seed=77, chunk=4)
    assert len(db) == n and os.path.exists(os.path.join(str(tmp_path), "models", "can", "points.xyz"))
    cfg.dataset.model_dir = os.path.join(str(tmp_path), "models")
    I.point_cloud_dict.clear()
    sym = deepIM_flownet()
    sym.get_symbol(cfg, is_train=True)
    params = sym.init_weights(cfg, {}, {}, seed=0)
    upd = batchUpdaterPyMulti(cfg, H, W, render_machine=rm)

    def run(private_copies):
        mod = MutableModule(cfg, params, B)
        loader = TrainDataLoader(None, db, cfg, batch_size=B, shuffle=False, device=DEV, workers=4, cache=PixelCache(DEV, 1 << 30))
        losses = []
        for epoch in range(2):
            loader.reset()
            for batch in loader:
                if private_copies:
                    batch = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in batch.items()}
                    torch.cuda.synchronize()
                outs = fit_batch(mod, batch, upd, 1e-4)
                losses.append(float(outs[-1]["flow_loss_sum"]) + float(outs[-1]["point_matching_loss_sum"]))
        hits = loader.cache.hits
        loader.close()
        return mod.flat_w.clone(), losses, hits

    w_a, loss_a, hits_a = run(False)
    w_b, loss_b, hits_b = run(True)
    assert torch.isfinite(w_a).all() and hits_a >= 5 * n   # the whole second epoch came from the cache
    # (the loss read-outs are sums of per-workgroup atomic adds: equal up to the order of the additions; the weights are not)
    np.testing.assert_allclose(loss_a, loss_b, rtol=1e-5)
    assert all(np.isfinite(loss_a)) and max(loss_a) > 0
    assert torch.equal(w_a, w_b)
    cfg.dataset.class_name = ["ape"]
    cfg.network.TRAIN_ITER_SIZE = 4
