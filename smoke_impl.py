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
    from loop_parity import check_loop, moving_head, oracle_free_and_forced
    from scene import make_scene, make_test_config

    cfg = make_test_config(test_iter=2)
    sym = deepIM_flownet()
    sym.get_symbol(cfg, is_train=False)
    params = sym.init_weights(cfg, {}, {}, seed=0)
    moving_head(params, seed=1)   # 3-12 deg / 4-42 mm per iteration, so that the check below guards the feedback (tests/loop_parity.py)
    scene = make_scene(B=1, seed=7, subdiv=3)
    pred = Predictor(cfg, params, 1)
    rm = Render_Py(None, cfg.dataset.class_name, scene["K"], meshes=scene["models"])
    ref = Refiner(cfg, pred, rm, 1)
    bl = scene["blobs"]
    ref.load(bl["image_observed"], bl["image_rendered"], bl["mask_observed"], bl["mask_rendered"], bl["src_pose"], bl["class_index"])
    poses = ref.refine().cpu().numpy()
    blobs = {k: bl[k][:1] for k in ("image_observed", "image_rendered", "mask_observed", "mask_rendered", "src_pose")}
    free, forced = oracle_free_and_forced(params, scene["models"][0], blobs, scene["K"], cfg.network.PIXEL_MEANS, poses[:, 0], test_iter=2)
    pts = scene["models"][0][0].astype(np.float64)
    rows = check_loop(bl["src_pose"][0], poses[:, 0], ref.se3_iter[:, 0].cpu().numpy(), free, forced, pts,
                      float(np.linalg.norm(pts.max(0) - pts.min(0))), tag="smoke", mean_rot_deg=2.0)
    err = max(r[2] for r in rows)
    print("smoke ok: 2-iteration refinement on cuda:0 matches the CPU oracle step by step ({:.1f} / {:.1f} deg per iteration), "
          "max |dpose_hip - dpose_oracle| = {:.2e}".format(rows[0][0], rows[1][0], err))
