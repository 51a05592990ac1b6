"""diagnose per-iteration flow/mask/zoom-factor differences of the full-graph refine loop vs the oracle"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "mx-deepim_amd"), os.path.join(ROOT, "tests")]
import numpy as np, torch
from scene import make_scene, make_test_config
from oracle import refine as orefine
from deepim.core.tester import Predictor, Refiner
from deepim.symbols.deepIM_flownet import deepIM_flownet
from lib.render_hip.render_py_multi import Render_Py

cfg = make_test_config(test_iter=4); cfg.TEST.FAST_TEST = False; cfg.dataset.class_name = ["ape", "can", "cat"]
sym = deepIM_flownet(); sym.get_symbol(cfg, is_train=False)
params = sym.init_weights(cfg, {}, {}, seed=3)
rng = np.random.RandomState(4)
params["trans_weight"] = (rng.randn(3, 256) * 0.002).astype(np.float32)
params["rot_weight"][1:] = (rng.randn(3, 256) * 0.01).astype(np.float32)
params["mask_conv3_weight"] = (rng.randn(1, 770, 3, 3) * 0.05).astype(np.float32)
params["mask_conv3_bias"] = np.array([0.1], np.float32)
B = 3
scene = make_scene(B=B, seed=909, subdiv=3, n_models=3); bl = scene["blobs"]
pred = Predictor(cfg, params, B)
rm = Render_Py(None, cfg.dataset.class_name, scene["K"], meshes=scene["models"])
ref = Refiner(cfg, pred, rm, B, capture_graph=False)
ref.load(bl["image_observed"], bl["image_rendered"], bl["mask_observed"], bl["mask_rendered"], bl["src_pose"], bl["class_index"])
zfs = []
orig = pred.net.forward_test
def fwd(*a, **k):
    o = orig(*a, **k); zfs.append(pred.net.zoom_factor.cpu().numpy().copy()); return o
pred.net.forward_test = fwd
poses = ref.refine().cpu().numpy().copy()
masks = ref.mask_pred_iter.cpu().numpy(); flows = ref.flow_est_iter.cpu().numpy()
for b in range(B):
    blobs_b = {k: bl[k][b:b + 1] for k in ("image_observed", "image_rendered", "mask_observed", "mask_rendered", "src_pose")}
    o_poses, o_se3, o_out = orefine.refine_pair(params, scene["models"][int(bl["class_index"][b])], blobs_b, scene["K"], cfg.network.PIXEL_MEANS,
                                                np.zeros(3), np.ones(3), "CAMERA", test_iter=4, fast_test=False, return_outputs=True)
    for it in range(4):
        rfl = o_out[it]["flow_est_crop"][0]
        d = np.abs(flows[it, b] - rfl)
        print("b", b, "it", it, "pose diff %.2e" % np.abs(poses[it, b] - o_poses[it]).max(), "zf", zfs[it][b], "ozf", o_out[it]["zoom_factor"][0],
              "flow max %.3f diff max %.4f n>tol %d" % (np.abs(rfl).max(), d.max(), (d > 1e-3 * max(1, np.abs(rfl).max())).sum()),
              "mask mism", (masks[it, b] != o_out[it]["mask_observed_pred"][0]).sum())
