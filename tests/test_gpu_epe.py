"""Test-time flow error on the device (reference deepim/core/tester.py:500-512, :675-736): dim_flow_epe_sums against the outputs of
the reference's own calc_EPE_one_pair (tests/golden/epe_golden.npz), and pred_eval's read-out on the full test graph against the
oracle restatement fed the package's calc_flow (itself pinned by the reference's calc_flow outputs, tests/test_data_layer.py)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import evaluation as oev  # noqa: E402
from scene import make_test_config, make_train_scene  # noqa: E402

DEV = "cuda:0"


def test_flow_epe_sums_vs_reference_outputs(hip_lib, golden_dir):
    from lib.hip import ops

    g = np.load(os.path.join(golden_dir, "flow_golden.npz"))
    e = np.load(os.path.join(golden_dir, "epe_golden.npz"))
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a, dtype=np.float32)).to(DEV)
    pred, flow = t(e["pred"].transpose(0, 3, 1, 2)), t(g["flow"].transpose(0, 3, 1, 2))
    vis, d0 = t(g["visible"][:, None]), t(g["depth_src"][:, None])
    sums = ops.flow_epe_sums(pred, flow, vis, d0)
    got = sums.cpu().numpy()
    want = e["out"]
    # float32 storage of calc_flow's float64 flow: <= 6e-8 relative per element; the sums themselves are float64 on both sides
    np.testing.assert_allclose(got[:, :3], want[:, [0, 2, 4]], rtol=3e-7)
    np.testing.assert_array_equal(got[:, 3:], want[:, [3, 5]])
    # accumulate adds, overwrite overwrites; sample order does not matter (per-sample sums); run-to-run bit-identical
    ops.flow_epe_sums(pred, flow, vis, d0, sums=sums, accumulate=True)
    np.testing.assert_allclose(sums.cpu().numpy(), 2 * got, rtol=1e-15)
    perm = torch.tensor([3, 0, 5, 1, 4, 2], device=DEV)
    again = ops.flow_epe_sums(pred[perm].contiguous(), flow[perm].contiguous(), vis[perm].contiguous(), d0[perm].contiguous())
    assert torch.equal(again.cpu(), torch.as_tensor(got)[perm.cpu()])
    # the float16 store is part of the definition (tester.py:485-487): a prediction already on the float16 grid gives the same sums,
    # one pushed off it by less than half a float16 step too
    p16 = pred.half().float()
    assert torch.equal(ops.flow_epe_sums(p16, flow, vis, d0).cpu(), torch.as_tensor(got))
    nudged = p16 * (1 + 1e-4)
    assert torch.equal(ops.flow_epe_sums(nudged, flow, vis, d0).cpu(), torch.as_tensor(got))
    # empty batch
    assert ops.flow_epe_sums(pred[:0], flow[:0], vis[:0], d0[:0]).shape == (0, 5)


def test_pred_eval_reports_flow_epe_on_the_full_graph(hip_lib):
    """BASELINE configs[3]'s shape (PRED_FLOW and not FAST_TEST): pred_eval scores the flow head's first output against calc_flow of the
    initial pair -- here against the oracle's calc_EPE_one_pair on the package's host calc_flow (float64, pinned by the reference's)."""
    from deepim.core.tester import Predictor, Refiner, pred_eval
    from deepim.symbols.deepIM_flownet import deepIM_flownet
    from lib.dataset.evaluation import PoseEvaluator
    from lib.pair_matching.flow import calc_flow
    from lib.render_hip.render_py_multi import Render_Py
    from oracle import native

    B = 2
    cfg = make_test_config(test_iter=2)
    cfg.TEST.FAST_TEST = False
    sym = deepIM_flownet()
    sym.get_symbol(cfg, is_train=False)
    params = sym.init_weights(cfg, {}, {}, seed=0)
    rng = np.random.RandomState(5)
    params["Convolution3_weight"] = (rng.randn(*params["Convolution3_weight"].shape) * 0.05).astype(np.float32)   # a flow head that says something
    scene = make_train_scene(B=B, seed=41, subdiv=3)
    bl = scene["blobs"]
    # the pair record's rendered depth (initial pose); depth_gt_observed is the render at the GT pose: zero off the object
    d_ren = np.stack([native.render(*scene["models"][0], scene["pose_init"][b][:, :3], scene["pose_init"][b][:, 3], scene["K"])[1][None]
                      for b in range(B)]).astype(np.float32)
    batch = {k: bl[k] for k in ("image_observed", "image_rendered", "mask_observed", "mask_rendered", "src_pose", "class_index")}
    batch.update(depth_rendered=d_ren, depth_gt_observed=bl["depth_gt_observed"], pose_observed=scene["pose_gt"].astype(np.float32))
    pred = Predictor(cfg, params, B)
    rm = Render_Py(None, cfg.dataset.class_name, scene["K"], meshes=scene["models"])
    ref = Refiner(cfg, pred, rm, B)
    pts = scene["models"][0][0].astype(np.float64)
    ev = PoseEvaluator(cfg.dataset.class_name, {cfg.dataset.class_name[0]: pts}, {cfg.dataset.class_name[0]: float(np.linalg.norm(pts.max(0) - pts.min(0)))})
    out = pred_eval(cfg, ref, [batch, batch], ev)
    flow_est = ref.flow_est_iter[0].cpu().numpy()
    assert np.abs(flow_est).max() > 0.5
    flows, viss = [], []
    for b in range(B):
        f, v, _ = calc_flow(d_ren[b, 0], bl["src_pose"][b], batch["pose_observed"][b], scene["K"], bl["depth_gt_observed"][b, 0],
                            standard_rep=False)
        flows.append(f.transpose(2, 0, 1)); viss.append(v[None])
    want = 2 * oev.epe_of_batch(flow_est, np.array(flows), np.array(viss), d_ren).sum(0)   # two identical batches
    e = out["epe"]
    got = np.array([e["sum_EPE_all"], e["sum_EPE_viz"], e["sum_EPE_vizbg"], e["num_inst_viz"], e["num_inst_vizbg"]])
    # counts: a pixel whose |dz| sits within float noise of the 3 mm visibility threshold may flip between the device labels (float32
    # flow, float64 predicate) and the host's -- none does on this scene; sums: float32 storage of the labels
    np.testing.assert_array_equal(got[3:], want[3:])
    np.testing.assert_allclose(got[:3], want[:3], rtol=2e-6)
    assert e["num_inst_all"] == 2 * B * 480 * 640 and 0 < e["num_inst_viz"] < e["num_inst_vizbg"] < e["num_inst_all"]
    np.testing.assert_allclose([e["epe_all"], e["epe_viz"], e["epe_vizbg"]],
                               [want[0] / e["num_inst_all"], want[1] / want[3], want[2] / want[4]], rtol=2e-6)
    # an undetected object (pose_rendered = -1, tester.py:451-475 leaves before the flow error) is not scored
    lost = dict(batch)
    lost["src_pose"] = np.array(bl["src_pose"], copy=True)
    lost["src_pose"][1] = -1.0
    e2 = pred_eval(cfg, ref, [lost], ev)["epe"]
    one = oev.epe_of_batch(flow_est[:1], np.array(flows[:1]), np.array(viss[:1]), d_ren[:1])[0]
    assert e2["num_inst_all"] == 480 * 640 and e2["num_inst_viz"] == one[3]
    np.testing.assert_allclose(e2["sum_EPE_all"], one[0], rtol=2e-6)
    # FAST_TEST: no flow output, no flow error
    cfg.TEST.FAST_TEST = True
    ref_fast = Refiner(cfg, Predictor(cfg, params, B), rm, B)
    assert "epe" not in pred_eval(cfg, ref_fast, [batch], ev)
