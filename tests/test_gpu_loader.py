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
    cfg.TEST.INIT_MASK = "mask_gt_observed"
    with pytest.raises(NotImplementedError):
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
    params["trans_weight"] = (np.random.RandomState(1).randn(3, 256) * 0.002).astype(np.float32)
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
        np.testing.assert_allclose(ref.refine().cpu().numpy(), direct[k], atol=1e-6)
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
