"""One small invocation of the hot path on cuda:0, checked against the CPU oracle (called by __graft_entry__.smoke)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "tests"))


def run():
    import torch

    assert torch.cuda.is_available(), "smoke() needs cuda:0"
    from deepim.core.tester import Predictor, Refiner
    from deepim.symbols.deepIM_flownet import deepIM_flownet
    from lib.render_hip.render_py_multi import Render_Py
    from oracle import refine as orefine
    from scene import make_scene, make_test_config

    cfg = make_test_config(test_iter=2)
    sym = deepIM_flownet()
    sym.get_symbol(cfg, is_train=False)
    params = sym.init_weights(cfg, {}, {}, seed=0)
    params["trans_weight"] = (np.random.RandomState(1).randn(3, 256) * 0.002).astype(np.float32)
    scene = make_scene(B=1, seed=7, subdiv=3)
    pred = Predictor(cfg, params, 1)
    rm = Render_Py(None, cfg.dataset.class_name, scene["K"], meshes=scene["models"])
    ref = Refiner(cfg, pred, rm, 1)
    bl = scene["blobs"]
    ref.load(bl["image_observed"], bl["image_rendered"], bl["mask_observed"], bl["mask_rendered"], bl["src_pose"], bl["class_index"])
    poses = ref.refine().cpu().numpy()
    blobs = {k: bl[k][:1] for k in ("image_observed", "image_rendered", "mask_observed", "mask_rendered", "src_pose")}
    o_poses, _ = orefine.refine_pair(params, scene["models"][0], blobs, scene["K"], cfg.network.PIXEL_MEANS, np.zeros(3), np.ones(3),
                                     "CAMERA", test_iter=2)
    err = max(np.abs(poses[i, 0] - o_poses[i]).max() for i in range(2))
    assert err < 1e-3, err
    print("smoke ok: 2-iteration refinement on cuda:0 matches the CPU oracle, max |dpose| = {:.2e}".format(err))
