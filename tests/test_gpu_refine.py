"""GPU parity of the network forward and of the whole 4-iteration refinement loop vs the CPU oracle."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import flownet as oflow, pose_error, refine as orefine  # noqa: E402
from scene import make_scene, make_test_config  # noqa: E402
from loop_parity import StaleRenderPose, check_loop, moving_head, oracle_free_and_forced, skip_box_mask  # noqa: E402

DEV = "cuda:0"


@pytest.fixture(scope="module")
def setup(hip_lib):
    assert torch.cuda.is_available()
    from deepim.symbols.deepIM_flownet import deepIM_flownet

    cfg = make_test_config(test_iter=4)
    sym = deepIM_flownet()
    sym.get_symbol(cfg, is_train=False)
    params = sym.init_weights(cfg, {}, {}, seed=0)
    # a pose head that moves the pose like a trained network does (3-12 deg, 4-42 mm per iteration: loop_parity.py); the
    # reference init (trans = 0, rot rows ~ U(0, 0.01)) moves it by 0.4 deg / 1-4 mm, below any bar that could guard the loop
    moving_head(params, seed=1)
    scene = make_scene(B=2, seed=2333, subdiv=3)
    return cfg, params, scene


def _pair_oracles(cfg, params, scene, b, poses_hip_b, test_iter=4, mesh=None, blobs=None, **kw):
    bl = scene["blobs"] if blobs is None else blobs
    keys = ("image_observed", "image_rendered", "mask_observed", "mask_rendered", "src_pose") + tuple(k for k in ("depth_observed", "depth_rendered") if k in bl)
    blobs_b = {k: bl[k][b:b + 1] for k in keys}
    mesh = scene["models"][int(bl["class_index"][b])] if mesh is None else mesh
    return oracle_free_and_forced(params, mesh, blobs_b, scene["K"], cfg.network.PIXEL_MEANS, poses_hip_b, test_iter=test_iter, **kw)


def test_forward_test_vs_oracle(setup):
    from deepim.core.tester import Predictor

    cfg, params, scene = setup
    B = 2
    pred = Predictor(cfg, params, B)
    batch = {k: torch.as_tensor(v).to(DEV) for k, v in scene["blobs"].items()}
    out = pred.predict(batch)[0]
    ref = oflow.forward_test(params, scene["blobs"], scene["K"], cfg.network.PIXEL_MEANS, fast_test=True)
    np.testing.assert_allclose(out["zoom_factor"].cpu().numpy(), ref["zoom_factor"], rtol=2e-6, atol=2e-6)
    X = pred.net.X.cpu().numpy().transpose(0, 3, 1, 2)
    # zoom_factor may differ in its last bit (np.dot(K, t) in float32 BLAS vs the kernel's left-to-right sum); at
    # x ~ 600 that moves a bilinear sample by up to ~1e-2 grey levels = 5e-5 after /255 (sampling itself is
    # pinned bit-tight with identical factors in test_gpu_ops.test_zoom_mask_and_image_vs_oracle)
    np.testing.assert_allclose(X[:, :6], ref["data"][:, :6], atol=1.5e-4)
    assert (X[:, 6:] != ref["data"][:, 6:]).sum() <= 16
    for name in ["flow_conv1", "conv3_1", "conv6_1"]:
        got = pred.net.acts[name].cpu().numpy().transpose(0, 3, 1, 2)
        want = ref["feats"][name].numpy()
        scale = np.abs(want).max()
        assert np.abs(got - want).max() <= 1e-4 * scale + 1e-5, name
    # north_star tolerance: outputs within 1e-3 fp32
    np.testing.assert_allclose(out["se3_output"].cpu().numpy(), ref["se3"], atol=1e-3)
    np.testing.assert_allclose(out["se3_output"].cpu().numpy(), ref["se3"], atol=2e-5, rtol=1e-4)


@pytest.mark.parametrize("graph", [False, True])
def test_refine_4iter_vs_oracle(setup, graph):
    from deepim.core.tester import Predictor, Refiner
    from lib.render_hip.render_py_multi import Render_Py

    cfg, params, scene = setup
    B = 2
    pred = Predictor(cfg, params, B)
    rm = Render_Py(None, cfg.dataset.class_name, scene["K"], meshes=scene["models"])
    ref = Refiner(cfg, pred, rm, B, capture_graph=graph)
    bl = scene["blobs"]
    ref.load(bl["image_observed"], bl["image_rendered"], bl["mask_observed"], bl["mask_rendered"], bl["src_pose"], bl["class_index"])
    poses = ref.refine().cpu().numpy().copy()
    poses2 = ref.refine().cpu().numpy()  # replay on the same batch must be idempotent
    np.testing.assert_array_equal(poses, poses2)
    assert int(ref.status_iter.abs().sum()) == 0
    pts = scene["models"][0][0].astype(np.float64)
    diam = np.linalg.norm(pts.max(0) - pts.min(0))
    se3 = ref.se3_iter.cpu().numpy()
    for b in range(B):
        free, forced = _pair_oracles(cfg, params, scene, b, poses[:, b])
        # the STEP of every iteration (3-12 deg, 4-42 mm here) against the oracle's step from the same state, at 1e-4 of
        # max(1, |step|); the free-running loops within 0.02 d of each other in ADD ("ADD(-S) vs reference" clause)
        check_loop(bl["src_pose"][b], poses[:, b], se3[:, b], free, forced, pts, diam, tag="graph={} pair {}".format(graph, b))


@pytest.mark.parametrize("fault", ["stale_render_pose", "skip_box_mask"])
def test_refine_loop_negative_controls(setup, fault, monkeypatch):
    """The bars of test_refine_4iter_vs_oracle must go RED when the loop is wrong: (1) iteration k rendered with the pose of
    iteration k-1, (2) mask_observed not rebuilt from the new rendered mask (data_pair.py:103-114 skipped).  Same scene, same
    checks; the faults are injected from outside (a wrapper around the render machine, a replaced ops.box_mask), eager mode."""
    from deepim.core.tester import Predictor, Refiner
    from lib.hip import ops
    from lib.render_hip.render_py_multi import Render_Py

    cfg, params, scene = setup
    B = 2
    bl = scene["blobs"]
    pred = Predictor(cfg, params, B)
    rm = Render_Py(None, cfg.dataset.class_name, scene["K"], meshes=scene["models"])
    if fault == "stale_render_pose":
        rm = StaleRenderPose(rm, torch.as_tensor(bl["src_pose"]).to(DEV))
    else:
        monkeypatch.setattr(ops, "box_mask", skip_box_mask(ops))
    ref = Refiner(cfg, pred, rm, B, capture_graph=False)
    ref.load(bl["image_observed"], bl["image_rendered"], bl["mask_observed"], bl["mask_rendered"], bl["src_pose"], bl["class_index"])
    poses = ref.refine().cpu().numpy().copy()
    se3 = ref.se3_iter.cpu().numpy()
    pts = scene["models"][0][0].astype(np.float64)
    diam = np.linalg.norm(pts.max(0) - pts.min(0))
    red = 0
    monkeypatch.undo()
    for b in range(B):
        free, forced = _pair_oracles(cfg, params, scene, b, poses[:, b])
        # iteration 0 sees the loaded blobs only: it must still agree (the fault is in the feedback, nothing else)
        np.testing.assert_allclose(poses[0, b], forced[0][0], atol=1e-5)
        try:
            check_loop(bl["src_pose"][b], poses[:, b], se3[:, b], free, forced, pts, diam, tag="{} pair {}".format(fault, b))
        except AssertionError as e:
            red += 1
            print("{} pair {}: red as required: {}".format(fault, b, str(e)[:200]))
    assert red == B, "the loop checks did not notice '{}'".format(fault)


def test_refine_full_graph_multiclass_vs_oracle(hip_lib):
    """BASELINE configs[3] shape at test size: several classes resident, FAST_TEST False (decoder + mask + flow heads
    produced every iteration, tester.py:485-491), UPDATE_MASK box_rendered (predicted mask not fed back), hipGraph."""
    from deepim.core.tester import Predictor, Refiner
    from deepim.symbols.deepIM_flownet import deepIM_flownet
    from lib.render_hip.render_py_multi import Render_Py

    cfg = make_test_config(test_iter=4)
    cfg.TEST.FAST_TEST = False
    cfg.dataset.class_name = ["ape", "can", "cat"]
    sym = deepIM_flownet()
    sym.get_symbol(cfg, is_train=False)
    params = sym.init_weights(cfg, {}, {}, seed=3)
    rng = np.random.RandomState(4)
    moving_head(params, seed=4)
    params["mask_conv3_weight"] = (rng.randn(1, 770, 3, 3) * 0.05).astype(np.float32)
    params["mask_conv3_bias"] = np.array([0.1], np.float32)
    B = 3
    scene = make_scene(B=B, seed=909, subdiv=3, n_models=3)
    bl = scene["blobs"]
    assert len(set(bl["class_index"].tolist())) >= 2
    pred = Predictor(cfg, params, B)
    rm = Render_Py(None, cfg.dataset.class_name, scene["K"], meshes=scene["models"])
    load = (bl["image_observed"], bl["image_rendered"], bl["mask_observed"], bl["mask_rendered"], bl["src_pose"], bl["class_index"])
    # eager pass with a hook that snapshots the blobs every iteration's forward sees
    ref = Refiner(cfg, pred, rm, B, capture_graph=False)
    ref.load(*load)
    snaps, orig = [], pred.net.forward_test

    def hooked(batch, **kw):
        snap = {k: v.cpu().numpy().copy() for k, v in batch.items()}
        if kw.get("src_pose") is not None:  # the loop hands the current pose over without copying it into the batch
            snap["src_pose"] = kw["src_pose"].cpu().numpy().copy()
        snaps.append(snap)
        return orig(batch, **kw)

    pred.net.forward_test = hooked
    poses = ref.refine().cpu().numpy().copy()
    pred.net.forward_test = orig
    masks = ref.mask_pred_iter.cpu().numpy().copy()
    flows = ref.flow_est_iter.cpu().numpy().copy()
    se3s = ref.se3_iter.cpu().numpy().copy()
    assert len(snaps) == 4
    # (1) every iteration's network outputs vs the oracle on the SAME blobs (tight: no error feedback through the loop)
    for it in range(4):
        o = oflow.forward_test(params, snaps[it], scene["K"], cfg.network.PIXEL_MEANS, fast_test=False)
        # the repo's step bar (tests/loop_parity.py): 2e-5 * max(1, |step|) -- the head's raw quaternion is ~8 long here, and an
        # element-wise rtol would hold its small components to 3e-6 of the vector, below what F(4x4,3x3) itself delivers
        bar = 2e-5 * max(1.0, float(np.abs(o["se3"]).max()))
        assert float(np.abs(se3s[it] - o["se3"]).max()) <= bar, (it, float(np.abs(se3s[it] - o["se3"]).max()), bar)
        assert (masks[it] != o["mask_observed_pred"]).sum() <= 300
        rfl = o["flow_est_crop"]
        np.testing.assert_allclose(flows[it], rfl, atol=1e-3 * max(1.0, np.abs(rfl).max()))
    # (2) the loop as a whole vs the oracle's loop (north_star bar 1e-3; random weights amplify pose differences
    #     through re-rendering, so later iterations of the dense heads are only compared in (1))
    for b in range(B):
        free, forced = _pair_oracles(cfg, params, scene, b, poses[:, b], fast_test=False, return_outputs=True)
        o_out = free[2]
        pts = scene["models"][int(bl["class_index"][b])][0].astype(np.float64)
        check_loop(bl["src_pose"][b], poses[:, b], se3s[:, b], free, forced, pts, np.linalg.norm(pts.max(0) - pts.min(0)),
                   tag="full graph pair {}".format(b))
        rfl = o_out[0]["flow_est_crop"][0]
        np.testing.assert_allclose(flows[0, b], rfl, atol=1e-3 * max(1.0, np.abs(rfl).max()))
        assert (masks[0, b] != o_out[0]["mask_observed_pred"][0]).sum() <= 100
    # (3) hipGraph replay of the same loop is bit-identical to the eager pass
    refg = Refiner(cfg, pred, rm, B, capture_graph=True)
    refg.load(*load)
    np.testing.assert_array_equal(refg.refine().cpu().numpy(), poses)
    np.testing.assert_array_equal(refg.mask_pred_iter.cpu().numpy(), masks)
    np.testing.assert_array_equal(refg.flow_est_iter.cpu().numpy(), flows)


def test_refine_modelnet_lit_vs_oracle(hip_lib):
    """BASELINE configs[4] shape at test size: several gray-textured meshes resident, lit renderer in the loop
    (tester.py:170-243), light intensities drawn from numpy's global RNG in the reference's order."""
    from deepim.core.tester import Predictor, Refiner
    from deepim.symbols.deepIM_flownet import deepIM_flownet
    from lib.render_hip.render_py_light_modelnet_multi import Render_Py_Light_ModelNet_Multi, vertex_normals

    cfg = make_test_config(test_iter=4)
    cfg.dataset.dataset = "ModelNet_v1"
    cfg.dataset.class_name = ["m0", "m1"]
    sym = deepIM_flownet()
    sym.get_symbol(cfg, is_train=False)
    params = sym.init_weights(cfg, {}, {}, seed=0)
    moving_head(params, seed=1)
    B = 2
    scene = make_scene(B=B, seed=4242, subdiv=3, n_models=2)
    bl = scene["blobs"]
    gray = np.full((32, 32, 3), 180, np.uint8)
    meshes = [(v, vertex_normals(v, f).astype(np.float32), t, f) for v, t, f, _ in scene["models"]]
    pred = Predictor(cfg, params, B)
    rm = Render_Py_Light_ModelNet_Multi(None, gray, scene["K"], 640, 480, 0.25, 6.0, brightness_ratios=[0.7], meshes=meshes)
    ref = Refiner(cfg, pred, rm, B, capture_graph=True)
    np.random.seed(99)
    ref.load(bl["image_observed"], bl["image_rendered"], bl["mask_observed"], bl["mask_rendered"], bl["src_pose"], bl["class_index"])
    poses = ref.refine().cpu().numpy().copy()
    assert int(ref.status_iter.abs().sum()) == 0
    np.random.seed(99)  # the oracle consumes the same stream: sample by sample, one draw per re-render
    for b in range(B):
        c = int(bl["class_index"][b])
        v, n, t, f = meshes[c]
        free, forced = _pair_oracles(cfg, params, scene, b, poses[:, b], mesh=(v, t, f, gray), lit={"normals": n, "ratio": 0.7})
        pts = v.astype(np.float64)
        check_loop(bl["src_pose"][b], poses[:, b], ref.se3_iter[:, b].cpu().numpy(), free, forced, pts,
                   np.linalg.norm(pts.max(0) - pts.min(0)), tag="modelnet lit pair {}".format(b))
    # the lit image really is what the loop fed back: shading varies over the object
    img = ref.batch["image_rendered"].cpu().numpy()
    on = ref.batch["mask_rendered"].cpu().numpy()[:, 0] > 0
    assert img[:, 0][on].std() > 2.0


def test_pred_eval_collects_and_scores(setup, tmp_path):
    """outer loop of pred_eval: result layout [cls][iter], result pickle, and the ADD(-S) clause of the metric -- our final poses
    scored against the oracle's final poses as 'ground truth' are within 0.02 d for every pair."""
    import pickle

    from deepim.core.tester import Predictor, Refiner, pred_eval
    from lib.dataset.evaluation import PoseEvaluator
    from lib.render_hip.render_py_multi import Render_Py

    _, params, scene = setup
    # this test is about the outer loop's bookkeeping, and it scores our poses against the FREE-RUNNING oracle as ground truth at
    # 0.1 deg / 1 mm: it keeps a head that moves the pose gently (the large steps are what test_refine_4iter_vs_oracle is for)
    params = dict(params)
    params["trans_weight"] = params["trans_weight"] * 0.1
    params["rot_weight"] = params["rot_weight"].copy()
    params["rot_weight"][1:] *= 0.05
    cfg = make_test_config(test_iter=4)  # the config object is a process-wide singleton that other tests re-shape
    B = 2
    pred = Predictor(cfg, params, B)
    rm = Render_Py(None, cfg.dataset.class_name, scene["K"], meshes=scene["models"])
    ref = Refiner(cfg, pred, rm, B, capture_graph=True)
    bl = scene["blobs"]
    z3, o3 = np.zeros(3), np.ones(3)
    oracle_final = []
    for b in range(B):
        blobs_b = {k: bl[k][b:b + 1] for k in ("image_observed", "image_rendered", "mask_observed", "mask_rendered", "src_pose")}
        o_poses, _ = orefine.refine_pair(params, scene["models"][int(bl["class_index"][b])], blobs_b, scene["K"], cfg.network.PIXEL_MEANS,
                                         z3, o3, "CAMERA", test_iter=4)
        oracle_final.append(o_poses[-1])
    batch = dict(bl)
    batch["pose_observed"] = np.array(oracle_final, dtype=np.float32)
    pts = scene["models"][0][0].astype(np.float64)
    diam = float(np.linalg.norm(pts.max(0) - pts.min(0)))
    ev = PoseEvaluator(cfg.dataset.class_name, {cfg.dataset.class_name[0]: pts}, {cfg.dataset.class_name[0]: diam})
    f = str(tmp_path / "results.pkl")
    out = pred_eval(cfg, ref, [batch, batch], ev, result_file=f)
    assert out["add"]["count_all"][0] == 4
    last = out["add"]["per_class"][(cfg.dataset.class_name[0], 3)]
    assert last["0.02"] == 100.0 and last["0.05"] == 100.0 and last["0.10"] == 100.0
    assert out["arp_2d"]["per_class"][(cfg.dataset.class_name[0], 3)]["2"] == 100.0
    assert max(out["all_rot_err"][0][3]) < 0.1 and max(out["all_trans_err"][0][3]) < 1e-3
    rot_err, trans_err, poses_est, poses_gt = pickle.load(open(f, "rb"))
    assert len(poses_est) == 1 and len(poses_est[0]) == 4 and len(poses_est[0][0]) == 4 and poses_est[0][0][0].shape == (3, 4)
    assert out["merged_over_ranks"] is False
    # "NO POINT VALID IN INIT POSE" (tester.py:419-445): pose_rendered = -1 everywhere marks an undetected object; it keeps its initial
    # pose and is scored 1000 deg / 1000 m at every iteration, the other pair of the batch is unaffected
    lost = dict(batch)
    lost["src_pose"] = np.array(bl["src_pose"], copy=True)
    lost["src_pose"][1] = -1.0
    out2 = pred_eval(cfg, ref, [lost], ev)
    assert out2["all_rot_err"][0][3] == [out["all_rot_err"][0][3][0], 1000] and out2["all_trans_err"][0][0][1] == 1000
    assert out2["add"]["per_class"][(cfg.dataset.class_name[0], 3)]["0.10"] == 50.0


def test_graph_variants_images_only_and_depth_input(hip_lib):
    """The other first-layer arities of get_convs (deepIM_flownet.py:33-66, :809-838): INPUT_MASK off -> ZoomImage derives the zoom window
    from the images and the network sees the 6 image channels; INPUT_DEPTH (without mask channels) adds the two zoomed depth planes.
    INPUT_DEPTH with the masks is the 10-channel first layer (two 8-lane groups on the device).  Forward outputs and a 2-iteration
    refinement against the oracle."""
    from deepim.core.tester import Predictor, Refiner
    from deepim.symbols.deepIM_flownet import deepIM_flownet, input_channels
    from lib.render_hip.render_py_multi import Render_Py

    B = 2
    scene = make_scene(B=B, seed=2333, subdiv=3)
    bl = scene["blobs"]
    z3, o3 = np.zeros(3), np.ones(3)
    rng = np.random.RandomState(3)
    depth = {"depth_observed": (rng.rand(B, 1, 480, 640) * 300).astype(np.float32), "depth_rendered": (rng.rand(B, 1, 480, 640) * 300).astype(np.float32)}
    for input_mask, pred_mask, input_depth in ((False, False, False), (False, True, False), (False, False, True), (True, False, True),
                                               (True, True, True)):
        cfg = make_test_config(test_iter=2)
        cfg.network.INPUT_MASK, cfg.network.PRED_MASK, cfg.network.INPUT_DEPTH = input_mask, pred_mask, input_depth
        cin = input_channels(cfg)
        assert cin == 6 + 2 * input_depth + 2 * (input_mask and pred_mask)
        sym = deepIM_flownet()
        sym.get_symbol(cfg, is_train=False)
        params = sym.init_weights(cfg, {}, {}, seed=4)
        assert params["flow_conv1_weight"].shape == (64, cin, 7, 7)
        moving_head(params, seed=5)
        if cin > 6:
            params["flow_conv1_weight"][:, 6:] = (rng.randn(64, cin - 6, 7, 7) * 0.05).astype(np.float32)   # the depth lanes must matter
        pred = Predictor(cfg, params, B)
        batch = {k: torch.as_tensor(np.ascontiguousarray(v)).to(DEV) for k, v in list(bl.items()) + list(depth.items())}
        out = pred.predict(batch)[0]
        kw = dict(input_mask=input_mask, pred_mask=pred_mask, input_depth=input_depth)
        host = dict(bl, **depth)
        ref = oflow.forward_test(params, host, scene["K"], cfg.network.PIXEL_MEANS, fast_test=True, **kw)
        np.testing.assert_allclose(out["zoom_factor"].cpu().numpy(), ref["zoom_factor"], atol=1e-5)
        # (the observed image sits on uniform noise and the depth planes here ARE noise: the steepest possible bilinear gradients, so an
        # ulp of the f32 sample coordinate shows as ~5e-5 here; smooth content agrees to 1e-5, tests/test_gpu_ops.py)
        np.testing.assert_allclose(pred.net.X[..., :min(cin, 8)].cpu().numpy(), ref["data"].transpose(0, 2, 3, 1)[..., :8], atol=2e-4)
        assert cin >= 8 or float(pred.net.X[..., 6:].abs().max()) == 0.0
        if cin == 10:   # the mask lanes live in the second 8-lane group
            np.testing.assert_array_equal(pred.net.X2[..., :2].cpu().numpy(), ref["data"].transpose(0, 2, 3, 1)[..., 8:])
            assert float(pred.net.X2[..., 2:].abs().max()) == 0.0
        # the repo's step bar, 2e-5 * max(1, |step|) (tests/loop_parity.py): the head's raw quaternion is ~10 long here
        se3_bar = 2e-5 * max(1.0, float(np.abs(ref["se3"]).max()))
        assert float(np.abs(out["se3_output"].cpu().numpy() - ref["se3"]).max()) <= se3_bar
        if True:   # the loop re-renders images, masks and -- INPUT_DEPTH -- the rendered depth plane
            rm = Render_Py(None, cfg.dataset.class_name, scene["K"], meshes=scene["models"])
            refiner = Refiner(cfg, pred, rm, B, capture_graph=True)
            dkw = dict(depth_observed=depth["depth_observed"], depth_rendered=depth["depth_rendered"]) if input_depth else {}
            refiner.load(bl["image_observed"], bl["image_rendered"], bl["mask_observed"], bl["mask_rendered"], bl["src_pose"], bl["class_index"],
                         **dkw)
            poses = refiner.refine().cpu().numpy()
            se3 = refiner.se3_iter.cpu().numpy()
            pts = scene["models"][0][0].astype(np.float64)
            for b in range(B):
                blobs = {k: host[k] for k in ("image_observed", "image_rendered", "mask_observed", "mask_rendered", "src_pose", "class_index")
                         + (("depth_observed", "depth_rendered") if input_depth else ())}
                free, forced = _pair_oracles(cfg, params, scene, b, poses[:, b], test_iter=2, blobs=blobs, **kw)
                check_loop(bl["src_pose"][b], poses[:, b], se3[:, b], free, forced, pts, np.linalg.norm(pts.max(0) - pts.min(0)),
                           tag="cin {} pair {}".format(cin, b), mean_rot_deg=2.0, min_rot_deg=0.5, min_trans_m=1e-3)
