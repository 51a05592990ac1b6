"""BASELINE configs[1] at full size -- the exact shape bench.py times: 16 pairs per GPU, 4 iterations, 480x640, FAST_TEST graph,
20 480-triangle mesh, hipGraph replay, a pose head that moves the pose 3-12 deg per iteration -- through size-independent properties
plus the oracle on four of the sixteen pairs (~0.4 s per pair-iteration on the host), every iteration's STEP teacher-forced
(tests/loop_parity.check_loop), the rest covered by invariances."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from loop_parity import check_loop, moving_head, oracle_free_and_forced  # noqa: E402
from scene import make_test_config  # noqa: E402

DEV = "cuda:0"


def test_batch16_properties(hip_lib):
    from deepim.core.tester import Predictor, Refiner
    from deepim.symbols.deepIM_flownet import deepIM_flownet
    from lib.render_hip.render_py_multi import Render_Py
    from lib.utils import synthetic as syn

    cfg = make_test_config(test_iter=4)
    sym = deepIM_flownet()
    sym.get_symbol(cfg, is_train=False)
    params = sym.init_weights(cfg, {}, {}, seed=0)
    moving_head(params, seed=1)    # the bench's head: 3-12 deg / 4-42 mm per iteration, every re-render covers new pixels
    B = 16
    models = syn.make_models(seed=2333, n_models=1, subdiv=5)    # 20 480 triangles, the bench's mesh
    assert models[0][2].shape[0] == 20480
    rm = Render_Py(None, cfg.dataset.class_name, syn.LINEMOD_K, meshes=models)
    batch = syn.build_device_batch(rm, B, seed=77)
    load = [batch[k] for k in ("image_observed", "image_rendered", "mask_observed", "mask_rendered", "src_pose", "class_index")]

    pred = Predictor(cfg, params, B)
    eager = Refiner(cfg, pred, rm, B, capture_graph=False)
    eager.load(*load)
    p_eager = eager.refine().cpu().numpy().copy()
    assert np.isfinite(p_eager).all() and int(eager.status_iter.abs().sum()) == 0

    # (1) hipGraph replay == eager, bit for bit, and replays are idempotent
    graph = Refiner(cfg, pred, rm, B, capture_graph=True)
    graph.load(*load)
    p_g1 = graph.refine().cpu().numpy().copy()
    p_g2 = graph.refine().cpu().numpy().copy()
    np.testing.assert_array_equal(p_g1, p_eager)
    np.testing.assert_array_equal(p_g2, p_g1)

    # (2) every refined pose is a rigid transform: R^T R = I, det = +1
    R = p_eager[..., :3].astype(np.float64)
    np.testing.assert_allclose(np.einsum("ibkj,ibkl->ibjl", R, R), np.broadcast_to(np.eye(3), R.shape), atol=1e-5)
    np.testing.assert_allclose(np.linalg.det(R), 1.0, atol=1e-5)

    # (3) samples are independent: permuting the batch permutes the result, and a 2-pair executor gives the same poses for pairs 3
    #     and 11.  Different tiles / split-K slabs see each pair, so equality is to f32 summation order -- asserted on the FIRST
    #     iteration (identical inputs; 1e-5 on poses that move 3-12 deg); under this head a 1e-6 difference grows 10-200x per
    #     iteration, so the later iterations are barred from identical state by (4), not here
    perm = np.random.RandomState(5).permutation(B)
    tperm = torch.as_tensor(perm, device=DEV)
    eager.load(*[t[tperm] for t in load])
    p_perm = eager.refine().cpu().numpy()
    np.testing.assert_allclose(p_perm[0], p_eager[0][perm], atol=1e-5)
    assert np.isfinite(p_perm).all()
    pred2 = Predictor(cfg, params, 2)
    small = Refiner(cfg, pred2, rm, 2, capture_graph=False)
    pick = torch.as_tensor([3, 11], device=DEV)
    small.load(*[t[pick] for t in load])
    np.testing.assert_allclose(small.refine().cpu().numpy()[0], p_eager[0][[3, 11]], atol=1e-5)

    # (4) the oracle on four of the sixteen pairs, against the hipGraph replay: every iteration's step from the same state within
    #     2e-5 * max(1, |step|), the scene really moves (>= 1.5 deg and 2 mm in every iteration), ADD from the same state < 0.02 d
    host = {k: batch[k].cpu().numpy() for k in ("image_observed", "image_rendered", "mask_observed", "mask_rendered", "src_pose")}
    se3 = graph.se3_iter.cpu().numpy()
    pts = models[0][0].astype(np.float64)
    diam = float(np.linalg.norm(pts.max(0) - pts.min(0)))
    for b in (0, 5, 9, 14):
        blobs_b = {k: v[b:b + 1] for k, v in host.items()}
        free, forced = oracle_free_and_forced(params, models[0], blobs_b, syn.LINEMOD_K, cfg.network.PIXEL_MEANS, p_g1[:, b], test_iter=4)
        check_loop(host["src_pose"][b], p_g1[:, b], se3[:, b], free, forced, pts, diam, tag="configs[1] pair {}".format(b))


def test_full_graph_batch16_and_many_meshes(hip_lib):
    """BASELINE configs[3] / [4] shapes: 16 pairs per GPU through the FULL test graph (decoder + mask + flow heads every iteration),
    classes drawn from 24 resident meshes.  Properties: graph == eager bit for bit, heads finite and non-trivial, two pairs vs the oracle."""
    from deepim.core.tester import Predictor, Refiner
    from deepim.symbols.deepIM_flownet import deepIM_flownet
    from lib.render_hip.render_py_multi import Render_Py
    from lib.utils import synthetic as syn

    cfg = make_test_config(test_iter=2)
    cfg.TEST.FAST_TEST = False
    n_cls = 24
    cfg.dataset.class_name = ["obj{:02d}".format(i) for i in range(n_cls)]
    sym = deepIM_flownet()
    sym.get_symbol(cfg, is_train=False)
    params = sym.init_weights(cfg, {}, {}, seed=2)
    rng = np.random.RandomState(3)
    moving_head(params, seed=3)
    params["mask_conv3_weight"] = (rng.randn(1, 770, 3, 3) * 0.05).astype(np.float32)
    B = 16
    models = syn.make_models(seed=77, n_models=n_cls, subdiv=2)
    rm = Render_Py(None, cfg.dataset.class_name, syn.LINEMOD_K, meshes=models)
    batch = syn.build_device_batch(rm, B, seed=5, n_classes=n_cls)
    assert len(set(batch["class_index"].cpu().numpy().tolist())) >= 8
    load = [batch[k] for k in ("image_observed", "image_rendered", "mask_observed", "mask_rendered", "src_pose", "class_index")]
    pred = Predictor(cfg, params, B)
    eager = Refiner(cfg, pred, rm, B, capture_graph=False)
    eager.load(*load)
    p_e = eager.refine().cpu().numpy().copy()
    m_e, f_e = eager.mask_pred_iter.cpu().numpy().copy(), eager.flow_est_iter.cpu().numpy().copy()
    graph = Refiner(cfg, pred, rm, B, capture_graph=True)
    graph.load(*load)
    np.testing.assert_array_equal(graph.refine().cpu().numpy(), p_e)
    np.testing.assert_array_equal(graph.mask_pred_iter.cpu().numpy(), m_e)
    np.testing.assert_array_equal(graph.flow_est_iter.cpu().numpy(), f_e)
    assert np.isfinite(f_e).all() and np.abs(f_e).max() > 0.1 and set(np.unique(m_e)) <= {0.0, 1.0} and 0.0 < m_e.mean() < 1.0
    host = {k: batch[k].cpu().numpy() for k in ("image_observed", "image_rendered", "mask_observed", "mask_rendered", "src_pose")}
    cls = batch["class_index"].cpu().numpy()
    se3 = eager.se3_iter.cpu().numpy()
    for b in (2, 13):
        blobs_b = {k: v[b:b + 1] for k, v in host.items()}
        mesh = models[int(cls[b])]
        free, forced = oracle_free_and_forced(params, mesh, blobs_b, syn.LINEMOD_K, cfg.network.PIXEL_MEANS, p_e[:, b], test_iter=2, fast_test=False)
        pts = mesh[0].astype(np.float64)
        check_loop(host["src_pose"][b], p_e[:, b], se3[:, b], free, forced, pts, float(np.linalg.norm(pts.max(0) - pts.min(0))),
                   tag="full graph pair {}".format(b))
    cfg.TEST.FAST_TEST = True
