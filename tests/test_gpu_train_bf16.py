"""bf16 training mode (BASELINE configs[2]: "bf16") of the training executor against the fp32 executor on the same batch.

The reference trains in fp32 only (deepim/train.py:338-414), so there is no reference behaviour to match: the bar is DECLARED here.
Every convolution / large deconvolution (forward, input gradient, weight gradient) rounds its two operands to bf16 (2^-9 relative
each, nearest even) and accumulates in f32; 10 encoder layers deep that compounds to ~5e-3 on the activations.  Measured on MI355X:
rot / trans within 3e-5, flow / mask heads 5e-3 .. 8e-3 L2-relative.  The GRADIENTS carry a second, larger term that is not arithmetic:
LeakyReLU' is discontinuous at 0, and a unit whose pre-activation lies within the forward noise (~0.5 % of its scale) of zero takes
slope 1 in one run and 0.1 in the other.  About 0.4 % of the units do, each changes its gradient by 90 % -> sqrt(0.004 * 0.81) ~ 6 % L2
on every tensor below the first such layer (the fp32-vs-f64 comparison of tests/test_gpu_train.py sees the same effect at 1e-3 because
its forward noise is 1e-6).  Either gradient is a valid sub-gradient of the same function at a point 2^-9 away.

  outputs    rot_est_norm, trans_est            |bf16 - f32| <= 1e-3
             flow_est_crop, mask_logit          L2-relative  <= 2e-2
  gradients  every learnable tensor             L2-relative  <= 1.2e-1, median over the tensors <= 6e-2, cosine >= 0.99
  update     one SGD step from the bf16 gradients moves every tensor within 15 % (L2) of the fp32 step
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from scene import make_train_config, make_train_scene  # noqa: E402

DEV = "cuda:0"


def l2rel(a, b):
    return float(np.linalg.norm((a - b).ravel()) / (np.linalg.norm(b.ravel()) + 1e-30))


def test_bf16_training_step_vs_fp32(hip_lib):
    from deepim.core.module import MutableModule
    from deepim.symbols.deepIM_flownet import deepIM_flownet

    cfg = make_train_config()
    sym = deepIM_flownet()
    sym.get_symbol(cfg, is_train=True)
    params = sym.init_weights(cfg, {}, {}, seed=0)
    rng = np.random.RandomState(1)
    params["trans_weight"] = (rng.randn(3, 256) * 0.002).astype(np.float32)
    params["rot_weight"][1:] = (rng.randn(3, 256) * 0.01).astype(np.float32)
    params["mask_conv3_weight"] = (rng.randn(1, 770, 3, 3) * 0.02).astype(np.float32)
    B = 2
    scene = make_train_scene(B=B, seed=99, subdiv=3)
    batch = {k: torch.as_tensor(np.ascontiguousarray(v)).to(DEV) for k, v in scene["blobs"].items()}
    m32 = MutableModule(cfg, params, B)
    m16 = MutableModule(cfg, params, B, compute_dtype="bf16")
    assert m16.net.bf16 and not m16.net.wino and m16.net.packed["conv3"].dtype == torch.bfloat16
    o32 = {k: v.clone() for k, v in m32.forward_backward(batch).items()}
    o16 = {k: v.clone() for k, v in m16.forward_backward(batch).items()}
    for k in ("rot_est_norm", "trans_est"):
        d = (o16[k] - o32[k]).abs().max().item()
        print("{:16s} max |bf16 - f32| = {:.2e}".format(k, d))
        assert d <= 1e-3, (k, d)
    for k in ("flow_est_crop", "mask_logit"):
        e = l2rel(o16[k].cpu().numpy(), o32[k].cpu().numpy())
        print("{:16s} L2-relative    = {:.2e}".format(k, e))
        assert e <= 2e-2, (k, e)
    g32, g16 = m32.get_grads(), m16.get_grads()
    errs = {}
    for k, a in g32.items():
        if np.abs(a).max() == 0:
            continue
        errs[k] = l2rel(g16[k], a)
        print("grad {:28s} L2-relative = {:.2e}".format(k, errs[k]))
    assert all(np.isfinite(v) for v in errs.values())
    assert max(errs.values()) <= 1.2e-1, max(errs.items(), key=lambda kv: kv[1])
    assert float(np.median(list(errs.values()))) <= 6e-2
    for k in errs:
        a, b = g16[k].ravel().astype(np.float64), g32[k].ravel().astype(np.float64)
        assert a @ b / (np.linalg.norm(a) * np.linalg.norm(b)) >= 0.99, k
    # one SGD step each
    before = m32.get_params()
    m32.update(cfg.TRAIN.lr)
    m16.force_bf16_bucket = True   # round the gradient bucket through bf16 as the multi-rank path does
    m16.update(cfg.TRAIN.lr)
    p32, p16 = m32.get_params(), m16.get_params()
    for k in before:
        step = p32[k] - before[k]
        if np.abs(step).max() == 0:
            np.testing.assert_array_equal(p16[k], before[k])   # frozen tensors
            continue
        assert l2rel(p16[k] - before[k], step) <= 0.15, k
    # the refreshed bf16 copies are what a fresh bf16 executor would build from the updated master weights
    o_next = m16.forward(batch)["rot_est_norm"].clone()
    fresh = MutableModule(cfg, p16, B, compute_dtype="bf16")
    np.testing.assert_array_equal(fresh.forward(batch)["rot_est_norm"].cpu().numpy(), o_next.cpu().numpy())


def test_bf16_gradients_with_f32_masks_and_bounded_flips(hip_lib):
    """Separates the two terms of the bf16-vs-f32 gradient difference.  (a) How many units take the other LeakyReLU branch: counted per
    layer, bounded at 1 %.  (b) Arithmetic only: the bf16 backward is run with every LeakyReLU' mask taken from the f32 executor
    (MutableModule.lrelu_mask_from, a test-only switch), so what remains is operand rounding -- bounded per tensor at 2e-2 L2-relative
    (a wrong tap / operand map in a bf16 gradient kernel shows as O(1) here; the 6e-2 .. 1.2e-1 bars of the un-masked comparison above
    could hide it)."""
    from deepim.core.module import MutableModule
    from deepim.symbols.deepIM_flownet import deepIM_flownet

    cfg = make_train_config()
    sym = deepIM_flownet()
    sym.get_symbol(cfg, is_train=True)
    params = sym.init_weights(cfg, {}, {}, seed=0)
    rng = np.random.RandomState(1)
    params["trans_weight"] = (rng.randn(3, 256) * 0.002).astype(np.float32)
    params["rot_weight"][1:] = (rng.randn(3, 256) * 0.01).astype(np.float32)
    params["mask_conv3_weight"] = (rng.randn(1, 770, 3, 3) * 0.02).astype(np.float32)
    B = 2
    scene = make_train_scene(B=B, seed=99, subdiv=3)
    batch = {k: torch.as_tensor(np.ascontiguousarray(v)).to(DEV) for k, v in scene["blobs"].items()}
    m32 = MutableModule(cfg, params, B)
    m16 = MutableModule(cfg, params, B, compute_dtype="bf16")
    m32.forward_backward(batch)
    g32 = m32.get_grads()
    m16.forward(batch)
    flips = m16.count_lrelu_flips(m32)
    total_f, total_n = 0, 0
    for k, (f, n) in flips.items():
        print("LeakyReLU branch flips {:12s} {:8d} / {:9d} = {:.3%}".format(k, f, n, f / n))
        assert f <= max(1e-2 * n, 8), (k, f, n)   # (fc6 / fc7 have 512 units: a handful of flips is already 1 %)
        total_f, total_n = total_f + f, total_n + n
    assert 0 < total_f < 5e-3 * total_n
    m16.lrelu_mask_from = m32
    m16.backward(batch)
    g16 = m16.get_grads()
    worst = 0.0
    for k, a in g32.items():
        if np.abs(a).max() == 0:
            continue
        e = l2rel(g16[k], a)
        worst = max(worst, e)
        print("grad (f32 masks) {:28s} L2-relative = {:.2e}".format(k, e))
        assert e <= 2e-2, (k, e)
    # the masks were the whole difference between the two bars: un-masked, the same tensors differ by several per cent
    m16.lrelu_mask_from = None
    m16.forward_backward(batch)
    g16_own = m16.get_grads()
    own = max(l2rel(g16_own[k], a) for k, a in g32.items() if np.abs(a).max() > 0)
    print("worst tensor: {:.2e} with f32 masks, {:.2e} with its own".format(worst, own))
    assert own > worst


@pytest.mark.parametrize("n_ranks", [4, 8])
def test_bf16_bucket_sum_over_n_ranks(hip_lib, n_ranks):
    """The multi-rank bf16 path sums the gradient buckets IN bf16 (the collective adds bf16 values: deepim/core/module.py
    _start_allreduce).  Emulated in one process: N rank gradients (one real gradient, N-1 scaled / perturbed copies, like pairs of
    one data set) are rounded to bf16 and added pairwise in bf16 in ring order, against the f32 sum: the error of the summed
    gradient stays below 2^-8 * sqrt(N) L2-relative per bucket and the SGD step it produces within 2 % of the step from the f32 sum."""
    from deepim.core.module import MutableModule
    from deepim.symbols.deepIM_flownet import deepIM_flownet
    from lib.hip import ops

    cfg = make_train_config()
    sym = deepIM_flownet()
    sym.get_symbol(cfg, is_train=True)
    params = sym.init_weights(cfg, {}, {}, seed=0)
    B = 2
    scene = make_train_scene(B=B, seed=99, subdiv=3)
    batch = {k: torch.as_tensor(np.ascontiguousarray(v)).to(DEV) for k, v in scene["blobs"].items()}
    mod = MutableModule(cfg, params, B, compute_dtype="bf16")
    mod.forward_backward(batch)
    g = mod.flat_g.clone()
    gen = torch.Generator(device=DEV)
    gen.manual_seed(3)
    ranks = [g * (1.0 + 0.3 * torch.randn(g.shape, generator=gen, device=DEV)) for _ in range(n_ranks)]
    exact = torch.stack(ranks).double().sum(0)
    acc = ops.to_bf16(ranks[0])
    for r in ranks[1:]:
        acc = (acc + ops.to_bf16(r)).to(torch.bfloat16)      # a bf16 add rounds its result to bf16, like the collective's reduction
    got = ops.from_bf16(acc).double()
    for a, b in mod.buckets:
        e = float((got[a:b] - exact[a:b]).norm() / exact[a:b].norm())
        print("bucket [{}, {}): bf16 ring sum over {} ranks, L2-relative error {:.2e}".format(a, b, n_ranks, e))
        assert e <= 2.0 ** -8 * np.sqrt(n_ranks), (a, b, e)
    lr, mom, wd = 1e-4, 0.975, 5e-4
    w0 = mod.flat_w.double()
    step_exact = -lr * (exact + wd * w0)
    step_bf16 = -lr * (got + wd * w0)
    nw = mod.n_weight
    assert float((step_bf16[:nw] - step_exact[:nw]).norm() / step_exact[:nw].norm()) <= 2e-2


def test_bf16_training_batch16_runs_and_stays_finite(hip_lib):
    """BASELINE configs[2] per-GPU size: 16 pairs, four chained optimizer steps in bf16 mode; the loss sums stay finite and the
    parameters move (throughput of this configuration: bench.py `train.bf16`)."""
    from deepim.core.module import MutableModule, fit_batch
    from deepim.symbols.deepIM_flownet import deepIM_flownet
    from lib.pair_matching.batch_updater_py_multi import batchUpdaterPyMulti
    from lib.render_hip.render_py_multi import Render_Py
    from lib.utils import synthetic as syn

    cfg = make_train_config()
    sym = deepIM_flownet()
    sym.get_symbol(cfg, is_train=True)
    params = sym.init_weights(cfg, {}, {}, seed=0)
    B = 16
    models = syn.make_models(seed=2333, n_models=1, subdiv=3)
    rm = Render_Py(None, cfg.dataset.class_name, syn.LINEMOD_K, meshes=models)
    batch = syn.build_device_train_batch(rm, B, seed=5, models=models)
    mod = MutableModule(cfg, params, B, compute_dtype="bf16")
    upd = batchUpdaterPyMulti(cfg, 480, 640, render_machine=rm)
    outs = fit_batch(mod, batch, upd, cfg.TRAIN.lr)
    assert len(outs) == 4 and mod.num_update == 4
    for o in outs:
        assert torch.isfinite(o["loss_sums"]).all()
    assert torch.isfinite(mod.flat_w).all()
    new = mod.get_params()
    assert np.abs(new["conv3_weight"] - params["conv3_weight"]).max() > 0
