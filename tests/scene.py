"""Seeded synthetic scene shared by CPU and GPU tests (inputs only; rendering by the CPU oracle)."""
import numpy as np

from lib.utils import synthetic as syn
from oracle import native


def make_scene(B=2, seed=2333, subdiv=3, n_models=1):
    """-> dict with models, K, blobs (numpy, reference blob names/shapes), pose_gt."""
    models = syn.make_models(seed=seed, n_models=n_models, subdiv=subdiv)
    cls, gt, init = syn.sample_pairs(seed + 1, B, n_classes=n_models)
    rng = np.random.default_rng(seed + 2)
    K = syn.LINEMOD_K
    H, W = 480, 640
    io, ir, mo, mr = [], [], [], []
    for b in range(B):
        v, t, f, tex = models[cls[b]]
        bgr_gt, d_gt = native.render(v, t, f, tex, gt[b][:, :3], gt[b][:, 3], K)
        obs = syn.compose_observed(bgr_gt, d_gt, rng)
        bgr_r, d_r = native.render(v, t, f, tex, init[b][:, :3], init[b][:, 3], K)
        io.append(syn.bgr_to_blob(obs))
        ir.append(syn.bgr_to_blob(bgr_r.astype(np.uint8)))
        m_r = (d_r > 0.2).astype(np.float32)
        mr.append(m_r[None, None])
        mo.append(syn.box_from_mask(m_r)[None, None])  # TEST.INIT_MASK == 'box_rendered' (image.py:437-460)
    blobs = {
        "image_observed": np.concatenate(io).astype(np.float32),
        "image_rendered": np.concatenate(ir).astype(np.float32),
        "mask_observed": np.concatenate(mo).astype(np.float32),
        "mask_rendered": np.concatenate(mr).astype(np.float32),
        "src_pose": init.astype(np.float32),
        "class_index": cls,
    }
    return {"models": models, "K": K, "blobs": blobs, "pose_gt": gt, "pose_init": init}


def make_test_config(test_iter=4):
    """the shipped ape test config, built from defaults + the YAML-level overrides (no file I/O)."""
    from deepim.config.config import config, reset_config

    reset_config()
    config.network.PIXEL_MEANS = np.array([123.68, 116.779, 103.939])
    config.network.INPUT_MASK = True
    config.network.PRED_MASK = True
    config.network.PRED_FLOW = True
    config.network.ROT_COORD = "CAMERA"
    config.dataset.NORMALIZE_FLOW = 20.0
    config.dataset.INTRINSIC_MATRIX = np.array([[572.4114, 0, 325.2611], [0, 573.57043, 242.04899], [0, 0, 1]], dtype=np.float32)
    config.dataset.trans_means = np.zeros(3, dtype=np.float32)
    config.dataset.trans_stds = np.ones(3, dtype=np.float32)
    config.dataset.class_name = ["ape"]
    config.TEST.test_iter = test_iter
    config.TEST.FAST_TEST = True
    config.TEST.UPDATE_MASK = "box_rendered"
    config.TEST.INIT_MASK = "box_rendered"
    return config


def make_train_scene(B=2, seed=2333, subdiv=3, npts=3000, n_models=1):
    """train-graph blobs: the test scene + mask_gt_observed, flow labels (depth->flow restatement), point clouds."""
    from oracle import se3 as ose3

    sc = make_scene(B=B, seed=seed, subdiv=subdiv, n_models=n_models)
    bl, K = sc["blobs"], sc["K"]
    rng = np.random.default_rng(seed + 5)
    mg, d_src, d_tgt, KT, pm, po = [], [], [], [], [], []
    for b in range(B):
        v, t, f, tex = sc["models"][int(bl["class_index"][b])]
        gt, init = sc["pose_gt"][b], sc["pose_init"][b]
        _, dg = native.render(v, t, f, tex, gt[:, :3], gt[:, 3], K)
        _, dr = native.render(v, t, f, tex, init[:, :3], init[:, 3], K)
        mg.append((dg > 0).astype(np.float32)[None, None])
        d_src.append(dr[None, None]); d_tgt.append(dg[None, None])
        R, tt = ose3.calc_se3(init, gt)
        KT.append(np.dot(K, np.concatenate([R, tt.reshape(3, 1)], axis=1)).astype(np.float32)[None])
        idx = rng.integers(0, v.shape[0], size=npts)
        P = v[idx].T.astype(np.float32)
        pm.append(P[None]); po.append((gt[:, :3] @ P + gt[:, 3:4]).astype(np.float32)[None])
    flow, valid = native.gpu_flow(np.concatenate(d_src), np.concatenate(d_tgt), np.concatenate(KT), np.linalg.inv(K).astype(np.float32))
    bl = dict(bl)
    bl["mask_gt_observed"] = np.concatenate(mg)
    bl["flow"] = flow
    bl["flow_weights"] = np.tile(valid, (1, 2, 1, 1))  # batch_updater_py_multi.py:352
    bl["point_cloud_model"] = np.concatenate(pm)
    bl["point_cloud_weights"] = np.ones_like(bl["point_cloud_model"])
    bl["point_cloud_observed"] = np.concatenate(po)
    bl["tgt_pose"] = sc["pose_gt"].astype(np.float32)
    bl["depth_gt_observed"] = np.concatenate(d_tgt).astype(np.float32)
    z3, o3 = np.zeros(3), np.ones(3)
    rt = [ose3.calc_RT_delta(sc["pose_init"][b], sc["pose_gt"][b], z3, o3, "CAMERA", "QUAT") for b in range(B)]
    bl["rot"] = np.stack([r for r, _ in rt]).astype(np.float32)
    bl["trans"] = np.stack([t for _, t in rt]).astype(np.float32)
    sc["blobs"] = bl
    return sc


def make_train_config():
    cfg = make_test_config(test_iter=4)
    cfg.network.TRAIN_ITER = True
    cfg.network.TRAIN_ITER_SIZE = 4
    cfg.train_iter.SE3_PM_LOSS = True
    cfg.train_iter.LW_PM = 0.1
    cfg.train_iter.NUM_3D_SAMPLE = 3000
    cfg.train_iter.LW_FLOW = 0.25
    cfg.train_iter.LW_MASK = 0.03
    cfg.TRAIN.lr = 0.0001
    cfg.TRAIN.momentum = 0.975
    cfg.TRAIN.wd = 0.0005
    return cfg
