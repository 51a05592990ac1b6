"""GPU parity of one full training step (forward + every gradient + SGD) vs the torch-CPU autograd oracle."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import train as otrain  # noqa: E402
from scene import make_train_config, make_train_scene  # noqa: E402

DEV = "cuda:0"


@pytest.fixture(scope="module")
def setup(hip_lib):
    assert torch.cuda.is_available()
    from deepim.symbols.deepIM_flownet import deepIM_flownet

    cfg = make_train_config()
    sym = deepIM_flownet()
    sym.get_symbol(cfg, is_train=True)
    params = sym.init_weights(cfg, {}, {}, seed=0)
    rng = np.random.RandomState(1)
    params["trans_weight"] = (rng.randn(3, 256) * 0.002).astype(np.float32)
    params["rot_weight"][1:] = (rng.randn(3, 256) * 0.01).astype(np.float32)
    params["mask_conv3_weight"] = (rng.randn(1, 770, 3, 3) * 0.02).astype(np.float32)
    scene = make_train_scene(B=2, seed=99, subdiv=3)
    return cfg, params, scene


def test_train_step_gradients_and_sgd(setup):
    from deepim.core.module import MutableModule

    cfg, params, scene = setup
    B = 2
    mod = MutableModule(cfg, params, B)
    batch = {k: torch.as_tensor(np.ascontiguousarray(v)).to(DEV) for k, v in scene["blobs"].items()}
    out = mod.forward_backward(batch)
    ref_out, ref_g = otrain.loss_and_grads(params, scene["blobs"], cfg, scene["K"])
    np.testing.assert_allclose(out["rot_est_norm"].cpu().numpy(), ref_out["rot_est_norm"], atol=1e-5)
    np.testing.assert_allclose(out["trans_est"].cpu().numpy(), ref_out["trans_est"], atol=1e-5)
    fe = ref_out["flow_est_crop"]
    np.testing.assert_allclose(out["flow_est_crop"].cpu().numpy(), fe, atol=1e-4 * max(1.0, np.abs(fe).max()))
    np.testing.assert_allclose(out["mask_logit"].cpu().numpy(), ref_out["mask_logit"], atol=1e-4)
    sums = mod.loss_sums.cpu().numpy()
    np.testing.assert_allclose(sums[0], ref_out["flow_loss_sum"], rtol=2e-3)
    np.testing.assert_allclose(sums[1], ref_out["pm_loss_sum"], rtol=2e-3)
    def rel(a, b):
        return np.abs(a - b).max() / (np.abs(b).max() + 1e-30)

    # LeakyReLU' is discontinuous at 0: a pre-activation within float32 accumulation noise of 0 (|x| < ~3e-5 here) takes slope 1 on
    # one side and 0.1 on the other, so a handful of elements per layer differ by 10x between this f32 run and the f64 oracle.
    # Evidence that this is the ONLY difference: with the GPU's own sign mask the deconv5 pre-activation gradient matches to 1e-5.
    c2 = mod.net.concat2.cpu().numpy().transpose(0, 3, 1, 2)[:, :1026]
    assert np.abs(c2 - ref_out["cat2"]).max() <= 1e-4 * np.abs(ref_out["cat2"]).max()
    flips = int(((c2[:, 512:1024] > 0) != (ref_out["cat2"][:, 512:1024] > 0)).sum())
    assert flips <= 20, flips
    dz5 = mod.dconcat2.cpu().numpy().transpose(0, 3, 1, 2)[:, 512:1024]
    assert rel(dz5, ref_out["d_cat2"][:, 512:1024] * np.where(c2[:, 512:1024] > 0, 1.0, 0.1)) <= 1e-5
    dc3 = mod.dconcat3.cpu().numpy().transpose(0, 3, 1, 2)
    dc2 = mod.dconcat2.cpu().numpy().transpose(0, 3, 1, 2)
    assert rel(dc3[:, :512], ref_out["d_cat3"][:, :512]) <= 1e-5 and rel(dc3[:, 768:770], ref_out["d_cat3"][:, 768:770]) <= 1e-5
    assert rel(dc2[:, :512], ref_out["d_cat2"][:, :512]) <= 1e-5 and rel(dc2[:, 1024:1026], ref_out["d_cat2"][:, 1024:1026]) <= 1e-5
    got = mod.get_grads()
    worst = {}
    for k, rg in ref_g.items():
        scale = np.abs(rg).max()
        err = np.abs(got[k] - rg).max()
        worst[k] = (err, scale)
        # relative to the tensor's largest gradient entry; sign(.) of the L1 point loss and the rounded zoom masks make a few
        # contributions flip between f32 and f64, hence 2e-3 rather than f32 epsilon
        l2 = np.linalg.norm((got[k] - rg).ravel()) / (np.linalg.norm(rg.ravel()) + 1e-30)
        print("grad {:28s} max|g| {:.3e}  max err {:.3e}  rel {:.2e}  l2 {:.2e}".format(k, scale, err, err / (scale + 1e-30), l2))
        exact_path = k.startswith(("fc", "rot", "trans", "Convolution", "deconv4", "upsample_flow", "mask_conv3"))  # no ReLU' flip upstream
        assert err <= (2e-5 if exact_path else 5e-2) * scale + 1e-9, (k, err, scale)
        assert l2 <= (2e-5 if exact_path else 1e-2), (k, l2)
    assert sum(1 for k, (e, s) in worst.items() if s > 0) >= 40  # every learnable tensor received a gradient
    # one SGD step
    moms = {k: np.zeros(v.shape) for k, v in params.items()}
    p2, moms = otrain.sgd_step({k: v.copy() for k, v in params.items()}, ref_g, moms, cfg.TRAIN.lr, cfg.TRAIN.momentum, cfg.TRAIN.wd)
    mod.update(cfg.TRAIN.lr)
    new = mod.get_params()
    for k in p2:
        step = np.abs(p2[k] - params[k]).max()
        assert np.abs(new[k] - p2[k]).max() <= 5e-2 * step + 1e-9, k
    # the refreshed packed weights give the same forward as a fresh executor built from the updated parameters
    out2 = mod.forward(batch)
    from deepim.core.module import MutableModule as MM

    mod2 = MM(cfg, new, B)
    out3 = mod2.forward(batch)
    np.testing.assert_allclose(out2["flow_est_crop"].cpu().numpy(), out3["flow_est_crop"].cpu().numpy(), atol=1e-6)
    np.testing.assert_array_equal(out2["rot_est_norm"].cpu().numpy(), out3["rot_est_norm"].cpu().numpy())


def test_batch_updater_and_fit_batch_vs_oracle(setup):
    """the between-iteration update (render -> labels -> depth->flow) and two chained optimizer steps"""
    from deepim.core.module import MutableModule, fit_batch
    from lib.pair_matching.batch_updater_py_multi import batchUpdaterPyMulti
    from lib.render_hip.render_py_multi import Render_Py
    from oracle import refine as orefine

    cfg, params, scene = setup
    B = 2
    bl = scene["blobs"]
    rm = Render_Py(None, cfg.dataset.class_name, scene["K"], meshes=scene["models"])
    upd = batchUpdaterPyMulti(cfg, 480, 640, render_machine=rm)
    batch = {k: torch.as_tensor(np.ascontiguousarray(v)).to(DEV) for k, v in bl.items()}
    preds = {"rot_est_norm": torch.tensor([[0.999, 0.02, -0.03, 0.01], [0.98, -0.1, 0.05, 0.12]], device=DEV),
             "trans_est": torch.tensor([[0.01, -0.02, 0.03], [-0.015, 0.01, -0.05]], device=DEV)}
    preds["rot_est_norm"] = preds["rot_est_norm"] / preds["rot_est_norm"].norm(dim=1, keepdim=True)
    new = upd.forward(batch, preds)
    z3, o3 = np.zeros(3), np.ones(3)
    ref = orefine.update_train_batch(bl, {"rot_est": preds["rot_est_norm"].cpu().numpy(), "trans_est": preds["trans_est"].cpu().numpy()},
                                     scene["models"], scene["K"], cfg.network.PIXEL_MEANS, z3, o3)
    np.testing.assert_allclose(new["src_pose"].cpu().numpy(), ref["src_pose"], atol=2e-6)
    np.testing.assert_allclose(new["rot"].cpu().numpy(), ref["rot"], atol=5e-6)
    np.testing.assert_allclose(new["trans"].cpu().numpy(), ref["trans"], atol=5e-6)
    assert (new["mask_rendered"].cpu().numpy() != ref["mask_rendered"]).sum() <= 8
    img_bad = (np.abs(new["image_rendered"].cpu().numpy() - ref["image_rendered"]).max(axis=1) > 1e-3).sum()
    assert img_bad <= 32, img_bad
    fw, rfw = new["flow_weights"].cpu().numpy(), ref["flow_weights"]
    assert (fw != rfw).sum() <= 200 and fw.sum() > 1000
    same = (fw == rfw) & (fw > 0)
    np.testing.assert_allclose(new["flow"].cpu().numpy()[same], ref["flow"][same], atol=2e-3)
    # two chained optimizer steps run and move the parameters; inputs of the 2nd step are the updater's outputs
    cfg.network.TRAIN_ITER_SIZE = 2
    mod = MutableModule(cfg, params, B)
    batch = {k: torch.as_tensor(np.ascontiguousarray(v)).to(DEV) for k, v in bl.items()}
    before = mod.flat_w.clone()
    outs = fit_batch(mod, batch, upd, cfg.TRAIN.lr)
    assert len(outs) == 2 and mod.num_update == 2
    assert torch.isfinite(mod.flat_w).all() and (mod.flat_w - before).abs().max() > 0
    assert not torch.equal(batch["src_pose"].cpu(), torch.as_tensor(bl["src_pose"]))  # src_pose advanced by the first iteration
    cfg.network.TRAIN_ITER_SIZE = 4


def test_adam_two_steps_and_optimizer_state_checkpoint(setup, tmp_path):
    """TRAIN.optimizer == 'adam' (train.py:338-375): two updates on the module's own gradients vs the float32 numpy restatement of
    mx.optimizer.Adam, then an optimizer-state save / load round trip into a fresh module."""
    from deepim.core.module import MutableModule

    cfg, params, scene = setup
    B = 2
    batch = {k: torch.as_tensor(np.ascontiguousarray(v)).to(DEV) for k, v in scene["blobs"].items()}
    old = cfg.TRAIN.optimizer
    cfg.TRAIN.optimizer = "adam"
    try:
        mod = MutableModule(cfg, params, B)
        p_ref = {k: v.copy() for k, v in params.items()}
        means = {k: np.zeros(v.shape, np.float32) for k, v in params.items()}
        var = {k: np.zeros(v.shape, np.float32) for k, v in params.items()}
        for t in (1, 2):
            mod.forward_backward(batch)
            g = mod.get_grads()
            p_ref, means, var = otrain.adam_step(p_ref, g, means, var, t, 1e-4, rescale_grad=1.0 / B)  # Module.init_optimizer: 1 / batch_size
            mod.update(1e-4)
            new = mod.get_params()
            for k in p_ref:
                if k in ("upsampling_weight", "mask_upsampling_weight"):
                    np.testing.assert_array_equal(new[k], params[k])  # frozen
                    continue
                # |step| ~ lr; a few ulp of the f32 division / sqrt on top
                assert np.abs(new[k] - p_ref[k]).max() <= 2e-3 * 1e-4 + 1e-9, (k, t)
            # the oracle continues from the module's parameters so that step 2 sees identical gradients
            p_ref = {k: v.copy() for k, v in new.items()}
        # the flat layout the gradient buckets and the vector kernels rely on: every tensor starts on a 128-byte boundary,
        # the buckets tile [0, n_weight + n_bias) exactly and in order, and the padding between tensors is still zero in the weights,
        # the gradients and both Adam states after two updates
        n_live = mod.n_weight + mod.n_bias
        assert mod.buckets[0][0] == 0 and mod.buckets[-1][1] == n_live
        assert all(b0[1] == b1[0] for b0, b1 in zip(mod.buckets, mod.buckets[1:])) and all(b > a and a % 32 == 0 for a, b in mod.buckets)
        used = torch.zeros(mod.flat_w.numel(), dtype=torch.bool, device=DEV)
        for n, off in mod.offset_of.items():
            assert off % 32 == 0, n
            sz = int(np.prod(mod.shapes[n]))
            assert not bool(used[off:off + sz].any()), n      # no two tensors overlap
            used[off:off + sz] = True
        gaps = ~used
        assert int(gaps.sum()) > 0     # the head + decoder + fc6 sizes are not multiples of 32: there IS padding to check
        for vec in (mod.flat_w, mod.flat_g, mod.flat_m, mod.flat_v):
            assert float(vec[gaps].abs().max()) == 0.0
        f = str(tmp_path / "opt.npz")
        mod.save_optimizer_states(f)
        mod2 = MutableModule(cfg, mod.get_params(), B)
        mod2.load_optimizer_states(f)
        assert mod2.num_update == 2
        np.testing.assert_array_equal(mod2.flat_m.cpu().numpy(), mod.flat_m.cpu().numpy())
        np.testing.assert_array_equal(mod2.flat_v.cpu().numpy(), mod.flat_v.cpu().numpy())
        mod.forward_backward(batch); mod.update(1e-4)
        mod2.forward_backward(batch); mod2.update(1e-4)
        np.testing.assert_array_equal(mod2.flat_w.cpu().numpy(), mod.flat_w.cpu().numpy())
    finally:
        cfg.TRAIN.optimizer = old


def test_batch_updater_multiclass(hip_lib):
    """BASELINE configs[2] shape at test size: several object classes in one training batch -- the between-iteration update renders
    every sample with ITS mesh and re-labels it (pose residual, flow, masks) like the per-sample loop of the reference does."""
    from lib.pair_matching.batch_updater_py_multi import batchUpdaterPyMulti
    from lib.render_hip.render_py_multi import Render_Py
    from oracle import refine as orefine

    cfg = make_train_config()
    cfg.dataset.class_name = ["ape", "can", "cat"]
    B = 3
    scene = make_train_scene(B=B, seed=4321, subdiv=3, n_models=3)
    bl = scene["blobs"]
    assert len(set(bl["class_index"].tolist())) >= 2
    rm = Render_Py(None, cfg.dataset.class_name, scene["K"], meshes=scene["models"])
    upd = batchUpdaterPyMulti(cfg, 480, 640, render_machine=rm)
    batch = {k: torch.as_tensor(np.ascontiguousarray(v)).to(DEV) for k, v in bl.items()}
    q = torch.tensor([[0.999, 0.02, -0.03, 0.01], [0.98, -0.1, 0.05, 0.12], [0.99, 0.05, 0.08, -0.06]], device=DEV)
    preds = {"rot_est_norm": q / q.norm(dim=1, keepdim=True),
             "trans_est": torch.tensor([[0.01, -0.02, 0.03], [-0.015, 0.01, -0.05], [0.02, 0.02, 0.01]], device=DEV)}
    new = upd.forward(batch, preds)
    z3, o3 = np.zeros(3), np.ones(3)
    ref = orefine.update_train_batch(bl, {"rot_est": preds["rot_est_norm"].cpu().numpy(), "trans_est": preds["trans_est"].cpu().numpy()},
                                     scene["models"], scene["K"], cfg.network.PIXEL_MEANS, z3, o3)
    np.testing.assert_allclose(new["src_pose"].cpu().numpy(), ref["src_pose"], atol=2e-6)
    np.testing.assert_allclose(new["rot"].cpu().numpy(), ref["rot"], atol=5e-6)
    np.testing.assert_allclose(new["trans"].cpu().numpy(), ref["trans"], atol=5e-6)
    assert (new["mask_rendered"].cpu().numpy() != ref["mask_rendered"]).sum() <= 12
    img_bad = (np.abs(new["image_rendered"].cpu().numpy() - ref["image_rendered"]).max(axis=1) > 1e-3).sum()
    assert img_bad <= 48, img_bad
    fw, rfw = new["flow_weights"].cpu().numpy(), ref["flow_weights"]
    assert (fw != rfw).sum() <= 300 and fw.sum() > 1000
    cfg.dataset.class_name = ["ape"]


@pytest.mark.parametrize("variant", [
    dict(SE3_PM_LOSS=True, SE3_PM_LOSS_TYPE="L2", SE3_DIST_LOSS=False),
    dict(SE3_PM_LOSS=True, SE3_PM_LOSS_TYPE="smooth_L1", SE3_PM_SL1_SCALAR=2.0, SE3_DIST_LOSS=False),
    dict(SE3_PM_LOSS=False, SE3_DIST_LOSS=True, LW_ROT=1.5, LW_TRANS=0.7, TRANS_LOSS_TYPE="L2"),
    dict(SE3_PM_LOSS=True, SE3_PM_LOSS_TYPE="L1", SE3_DIST_LOSS=True, LW_ROT=0.8, LW_TRANS=1.3, TRANS_LOSS_TYPE="smooth_L1",
         TRANS_SMOOTH_L1_SCALAR=3.0),
    dict(SE3_PM_LOSS=False, SE3_DIST_LOSS=True, LW_ROT=1.0, LW_TRANS=1.0, TRANS_LOSS_TYPE="L1"),
])
def test_pose_loss_variants(setup, variant):
    """The pose losses the shipped YAMLs leave off (deepIM_flownet.py:396-437, :458-499): SE3_DIST_LOSS (rot_loss + trans_loss with
    TRANS_LOSS_TYPE L2 / smooth_L1 / L1) and SE3_PM_LOSS_TYPE L2 / smooth_L1 -- loss sums and every gradient that does not pass a
    LeakyReLU' flip (pose head, fc6) against torch autograd of the oracle; the flip-exposed tensors keep the bar of the main test."""
    import copy

    from deepim.core.module import MutableModule

    cfg0, params, scene = setup
    cfg = copy.deepcopy(cfg0)
    for k, v in variant.items():
        cfg.train_iter[k] = v
    B = 2
    mod = MutableModule(cfg, params, B)
    batch = {k: torch.as_tensor(np.ascontiguousarray(v)).to(DEV) for k, v in scene["blobs"].items()}
    mod.forward_backward(batch)
    ref_out, ref_g = otrain.loss_and_grads(params, scene["blobs"], cfg, scene["K"])
    sums = mod.loss_sums.cpu().numpy()
    if cfg.train_iter.SE3_PM_LOSS:
        np.testing.assert_allclose(sums[1], ref_out["pm_loss_sum"], rtol=2e-3)
    if cfg.train_iter.SE3_DIST_LOSS:
        np.testing.assert_allclose(sums[3], ref_out["rot_loss_sum"], rtol=1e-4, atol=1e-6)
        np.testing.assert_allclose(sums[4], ref_out["trans_loss_sum"], rtol=1e-4, atol=1e-7)
    got = mod.get_grads()
    for k, rg in ref_g.items():
        scale = np.abs(rg).max()
        err = np.abs(got[k] - rg).max()
        exact_path = k.startswith(("fc", "rot", "trans", "Convolution", "deconv4", "upsample_flow", "mask_conv3"))
        # exact paths: f32 against f64 through fc6 (81920 products per output) with gradients 100x smaller than in the main test when only
        # the distance loss drives the head (measured 4e-5 on fc6_weight)
        assert err <= (1e-4 if exact_path else 5e-2) * scale + 1e-9, (variant, k, err, scale)
    for k in ("rot_weight", "trans_weight", "fc7_weight"):
        assert np.abs(ref_g[k]).max() > 0, k   # the variant really drives the pose head


@pytest.mark.parametrize("input_mask,input_depth", [(False, False), (False, True), (True, True)])
def test_train_input_arities(setup, input_mask, input_depth):
    """The first-layer input arities of get_convs (reference deepIM_flownet.py:33-66) in the TRAINING executor: 6 channels (images only:
    INPUT_MASK off, the zoom window still comes from the masks, :589-612), 8 (images + depth planes), 10 (+ masks, two 8-lane groups).
    Forward outputs and every gradient -- flow_conv1's in its (64, cin, 7, 7) layout -- against torch autograd of the oracle; one SGD
    step + repack gives the same forward as a fresh executor built from the updated parameters."""
    from deepim.core.module import MutableModule
    from deepim.symbols.deepIM_flownet import deepIM_flownet, input_channels

    cfg0, params0, scene = setup
    cfg = make_train_config()
    cfg.network.INPUT_MASK, cfg.network.INPUT_DEPTH = input_mask, input_depth
    cin = input_channels(cfg)
    assert cin == {(False, False): 6, (False, True): 8, (True, True): 10}[(input_mask, input_depth)]
    rng = np.random.RandomState(7)
    params = dict(params0)
    w1 = params0["flow_conv1_weight"]
    params["flow_conv1_weight"] = (rng.randn(64, cin, 7, 7) * float(w1.std())).astype(np.float32)
    B = 2
    blobs = dict(scene["blobs"])
    if input_depth:
        blobs["depth_observed"] = (rng.rand(B, 1, 480, 640) * 300).astype(np.float32)
        blobs["depth_rendered"] = (rng.rand(B, 1, 480, 640) * 300).astype(np.float32)
    mod = MutableModule(cfg, params, B)
    assert tuple(mod.w["flow_conv1_weight"].shape) == (64, cin, 7, 7)
    batch = {k: torch.as_tensor(np.ascontiguousarray(v)).to(DEV) for k, v in blobs.items()}
    out = mod.forward_backward(batch)
    ref_out, ref_g = otrain.loss_and_grads(params, blobs, cfg, scene["K"])
    np.testing.assert_allclose(out["rot_est_norm"].cpu().numpy(), ref_out["rot_est_norm"], atol=1e-5)
    np.testing.assert_allclose(out["trans_est"].cpu().numpy(), ref_out["trans_est"], atol=1e-5)
    np.testing.assert_allclose(out["mask_logit"].cpu().numpy(), ref_out["mask_logit"], atol=1e-4)
    got = mod.get_grads()
    for k, rg in ref_g.items():
        scale = np.abs(rg).max()
        l2 = np.linalg.norm((got[k] - rg).ravel()) / (np.linalg.norm(rg.ravel()) + 1e-30)
        exact_path = k.startswith(("fc", "rot", "trans", "Convolution", "deconv4", "upsample_flow", "mask_conv3"))
        # same bars as test_train_step_gradients_and_sgd (LeakyReLU' flips of pre-activations within f32 noise of zero below the decoder)
        err = np.abs(got[k] - rg).max()
        print("grad {:28s} max|g| {:.3e}  max err {:.3e}  l2 {:.2e}".format(k, scale, err, l2))
        # (a single LeakyReLU' flip in deconv5's output moves one entry of deconv4's gradient by 1.05e-4 of the largest, of
        # Convolution1's bias gradient by 1.3e-3: max bar 5e-3)
        assert err <= (5e-3 if exact_path else 5e-2) * scale + 1e-9, (k, float(err), float(scale))
        # "exact" = few LeakyReLU' sites upstream, not none: Convolution1's gradient passes deconv4's activation, and with the random
        # first-layer weights of this test one flip there shows as 1.2e-3 L2 (10-channel case); 1e-4 holds for the shipped weights
        assert l2 <= (2e-3 if exact_path else 1e-2), (k, l2)
    g1 = got["flow_conv1_weight"]
    assert g1.shape == (64, cin, 7, 7) and all(np.abs(g1[:, c]).max() > 0 for c in range(cin))   # every input channel got its gradient
    mod.update(cfg.TRAIN.lr)
    new = mod.get_params()
    assert np.abs(new["flow_conv1_weight"] - params["flow_conv1_weight"]).max() > 0
    out2 = mod.forward(batch)
    out3 = MutableModule(cfg, new, B).forward(batch)
    np.testing.assert_array_equal(out2["rot_est_norm"].cpu().numpy(), out3["rot_est_norm"].cpu().numpy())
    np.testing.assert_allclose(out2["flow_est_crop"].cpu().numpy(), out3["flow_est_crop"].cpu().numpy(), atol=1e-6)


@pytest.mark.parametrize("pred_flow,pred_mask,input_mask", [(True, False, True), (False, True, True), (False, False, True), (False, False, False)])
def test_train_optional_heads(setup, pred_flow, pred_mask, input_mask):
    """Training graphs without the flow and / or the mask head (reference: the decoder exists when either is predicted,
    deepIM_flownet.py:213; flow loss :315-357, mask loss :502-536; the zoom window comes from the masks when INPUT_MASK or PRED_MASK,
    else from the images, :589-640).  Outputs and every gradient of the parameters that configuration has, vs the oracle."""
    from deepim.core.module import MutableModule
    from deepim.symbols.deepIM_flownet import deepIM_flownet

    _, _, scene = setup
    cfg = make_train_config()
    cfg.network.PRED_FLOW, cfg.network.PRED_MASK, cfg.network.INPUT_MASK = pred_flow, pred_mask, input_mask
    sym = deepIM_flownet()
    sym.get_symbol(cfg, is_train=True)
    params = sym.init_weights(cfg, {}, {}, seed=0)
    rng = np.random.RandomState(1)
    params["trans_weight"] = (rng.randn(3, 256) * 0.002).astype(np.float32)
    params["rot_weight"][1:] = (rng.randn(3, 256) * 0.01).astype(np.float32)
    has_dec = pred_flow or pred_mask
    assert ("deconv5_weight" in params) == has_dec and ("Convolution3_weight" in params) == pred_flow and ("mask_conv3_weight" in params) == pred_mask
    B = 2
    blobs = scene["blobs"]
    mod = MutableModule(cfg, params, B)
    batch = {k: torch.as_tensor(np.ascontiguousarray(v)).to(DEV) for k, v in blobs.items()}
    out = mod.forward_backward(batch)
    ref_out, ref_g = otrain.loss_and_grads(params, blobs, cfg, scene["K"])
    np.testing.assert_allclose(out["zoom_factor"].cpu().numpy(), ref_out["zoom_factor"], atol=1e-5)
    np.testing.assert_allclose(out["rot_est_norm"].cpu().numpy(), ref_out["rot_est_norm"], atol=1e-5)
    np.testing.assert_allclose(out["trans_est"].cpu().numpy(), ref_out["trans_est"], atol=1e-5)
    assert ("flow_est_crop" in out) == pred_flow and ("mask_logit" in out) == pred_mask
    if pred_flow:
        fe = ref_out["flow_est_crop"]
        np.testing.assert_allclose(out["flow_est_crop"].cpu().numpy(), fe, atol=1e-4 * max(1.0, np.abs(fe).max()))
    if pred_mask:
        np.testing.assert_allclose(out["mask_logit"].cpu().numpy(), ref_out["mask_logit"], atol=1e-4)
    got = mod.get_grads()
    assert sorted(got) == sorted(ref_g)
    for k, rg in ref_g.items():
        scale = np.abs(rg).max()
        err = np.abs(got[k] - rg).max()
        l2 = np.linalg.norm((got[k] - rg).ravel()) / (np.linalg.norm(rg.ravel()) + 1e-30)
        print("grad {:28s} max|g| {:.3e}  max err {:.3e}  l2 {:.2e}".format(k, scale, err, l2))
        # no LeakyReLU' below these: the pose head (fc6 / fc7 activations aside) and the two heads that sit on Concat3; every other
        # gradient passes the decoder / encoder activations, where pre-activations within f32 noise of zero flip between f32 and f64
        exact_path = k.startswith(("fc", "rot", "trans", "Convolution3", "mask_conv3"))
        assert err <= (1e-4 if exact_path else 5e-2) * scale + 1e-9, (k, float(err), float(scale))
        assert l2 <= (1e-4 if exact_path else 1e-2), (k, l2)
    mod.update(cfg.TRAIN.lr)
    out2 = mod.forward(batch)
    out3 = MutableModule(cfg, mod.get_params(), B).forward(batch)
    np.testing.assert_array_equal(out2["rot_est_norm"].cpu().numpy(), out3["rot_est_norm"].cpu().numpy())


def test_batch_updater_modelnet_lit(setup):
    """The ModelNet branch of batchUpdaterPyMulti.forward (reference batch_updater_py_multi.py:232-270): the re-render between training
    iterations goes through the lit renderer, light index 2 moved by the refined translation, one host-drawn intensity per sample in
    batch order (numpy's global stream, like the reference)."""
    from lib.pair_matching.batch_updater_py_multi import batchUpdaterPyMulti
    from lib.render_hip.render_py_light_modelnet_multi import Render_Py_Light_ModelNet_Multi, vertex_normals
    from lib.render_hip.render_py_multi import Render_Py
    from oracle import refine as orefine

    cfg = make_train_config()
    cfg.dataset.dataset = "ModelNet_v1"
    cfg.dataset.class_name = ["m0", "m1"]
    B = 2
    scene = make_train_scene(B=B, seed=4242, subdiv=3, n_models=2)
    bl = scene["blobs"]
    gray = np.full((32, 32, 3), 180, np.uint8)
    meshes = [(v, vertex_normals(v, f).astype(np.float32), t, f) for v, t, f, _ in scene["models"]]
    rm = Render_Py_Light_ModelNet_Multi(None, gray, scene["K"], 640, 480, 0.25, 6.0, brightness_ratios=[0.7], meshes=meshes)
    with pytest.raises(Exception):   # a ModelNet batch must not be re-rendered unlit
        batchUpdaterPyMulti(cfg, 480, 640, render_machine=Render_Py(None, cfg.dataset.class_name, scene["K"], meshes=scene["models"]))
    upd = batchUpdaterPyMulti(cfg, 480, 640, render_machine=rm)
    batch = {k: torch.as_tensor(np.ascontiguousarray(v)).to(DEV) for k, v in bl.items()}
    preds = {"rot_est_norm": torch.tensor([[0.999, 0.02, -0.03, 0.01], [0.98, -0.1, 0.05, 0.12]], device=DEV),
             "trans_est": torch.tensor([[0.01, -0.02, 0.03], [-0.015, 0.01, -0.05]], device=DEV)}
    preds["rot_est_norm"] = preds["rot_est_norm"] / preds["rot_est_norm"].norm(dim=1, keepdim=True)
    np.random.seed(5)
    new = upd.forward(batch, preds)
    np.random.seed(5)
    z3, o3 = np.zeros(3), np.ones(3)
    ref = orefine.update_train_batch(bl, {"rot_est": preds["rot_est_norm"].cpu().numpy(), "trans_est": preds["trans_est"].cpu().numpy()},
                                     [(v, t, f, gray) for v, n, t, f in meshes], scene["K"], cfg.network.PIXEL_MEANS, z3, o3,
                                     lit={"normals": [n for v, n, t, f in meshes], "ratio": 0.7})
    np.testing.assert_allclose(new["src_pose"].cpu().numpy(), ref["src_pose"], atol=2e-6)
    np.testing.assert_allclose(new["rot"].cpu().numpy(), ref["rot"], atol=5e-6)
    assert (new["mask_rendered"].cpu().numpy() != ref["mask_rendered"]).sum() <= 8
    img, rimg = new["image_rendered"].cpu().numpy(), ref["image_rendered"]
    # lit shading: integral grey levels, a rounding tie may land one level apart; silhouette pixels may differ by the fill rule
    assert (np.abs(img - rimg).max(axis=1) > 1.0).sum() <= 32
    on = ref["mask_rendered"][:, 0] > 0
    assert rimg[:, 0][on].std() > 2.0 and np.abs(img - rimg)[:, :, :, :][np.broadcast_to(on[:, None], img.shape)].mean() < 0.2
    fw, rfw = new["flow_weights"].cpu().numpy(), ref["flow_weights"]
    assert (fw != rfw).sum() <= 200 and fw.sum() > 1000
