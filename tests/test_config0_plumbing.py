"""BASELINE configs[0] -- 'single ape pair, 1 iteration, CPU' (SURVEY 8d): the YAML goes through update_config, the symbol builds the
parameter table, seeded weights are created, and ONE iteration of the CPU restatement runs on ONE synthetic 480x640 pair.
Checks names / shapes / finiteness of everything the refinement loop reads.  No GPU, no HIP library."""
import os

import numpy as np

from oracle import flownet as oflow, refine as orefine
from scene import make_scene

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHIPPED = "/root/reference/experiments/deepim/cfgs/deepim_flownet_LM_SIXD_v1_ape_RFMx4_8epoch.yaml"
OURS = os.path.join(ROOT, "mx-deepim_amd", "experiments", "deepim", "cfgs", "deepim_hip_LM_ape_test.yaml")


def test_single_pair_one_iteration_cpu():
    from deepim.config.config import config, reset_config, update_config
    from deepim.symbols.deepIM_flownet import deepIM_flownet, input_channels

    reset_config()
    update_config(SHIPPED if os.path.exists(SHIPPED) else OURS)   # the reference's own YAML when it is present (build container)
    assert config.symbol == "deepIM_flownet" and config.dataset.class_name == ["ape"] and int(config.TEST.test_iter) == 4
    assert config.network.INPUT_MASK and config.network.PRED_MASK and config.network.PRED_FLOW and config.network.ROT_COORD == "CAMERA"
    assert input_channels(config) == 8
    sym = deepIM_flownet()
    sym.get_symbol(config, is_train=False)
    shapes = sym.infer_param_shapes(config)
    assert shapes["flow_conv1_weight"] == (64, 8, 7, 7) and shapes["fc6_weight"] == (256, 81920) and shapes["rot_weight"] == (4, 256)
    params = sym.init_weights(config, {}, {}, seed=0)
    assert set(params) == set(shapes) and all(params[k].shape == tuple(shapes[k]) and params[k].dtype == np.float32 for k in shapes)

    scene = make_scene(B=1, seed=11, subdiv=3)
    bl = scene["blobs"]
    assert bl["image_observed"].shape == (1, 3, 480, 640) and bl["mask_rendered"].shape == (1, 1, 480, 640) and bl["src_pose"].shape == (1, 3, 4)
    out = oflow.forward_test(params, bl, scene["K"], config.network.PIXEL_MEANS, fast_test=True)
    assert out["se3"].shape == (1, 7) and out["zoom_factor"].shape == (1, 4) and out["data"].shape == (1, 8, 480, 640)
    assert np.isfinite(out["se3"]).all() and out["zoom_factor"][0, 0] > 0
    blobs = {k: bl[k] for k in ("image_observed", "image_rendered", "mask_observed", "mask_rendered", "src_pose")}
    poses, se3s = orefine.refine_pair(params, scene["models"][0], blobs, scene["K"], config.network.PIXEL_MEANS, config.dataset.trans_means,
                                      config.dataset.trans_stds, config.network.ROT_COORD, test_iter=1)
    assert len(poses) == 1 and poses[0].shape == (3, 4) and np.isfinite(poses[0]).all()
    R = poses[0][:, :3]
    np.testing.assert_allclose(R.T @ R, np.eye(3), atol=1e-6)   # src_pose enters as float32
    reset_config()
