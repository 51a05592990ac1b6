"""Behavioural end-to-end check -- the reference's own integration test, "train, test, read the error fall over the iterations"
(/root/reference/train_and_test_deepim_ape.sh, README.md:167-181, LM6D_REFINE.py:461-681): the system LEARNS and the refinement loop
CONTRACTS, and under those weights the free-running HIP loop and the oracle loop stay together over all four iterations.

  1. train from init_weights on a fixed set of 32 synthetic single-object pairs through fit_batch (TRAIN_ITER_SIZE 4: forward, all
     losses, backward, update, then re-render + re-label between the inner iterations -- reference deepim/core/module.py:1205-1213),
     TRAIN.optimizer adam (train.py:338-375), 250 epochs at 1e-4 (29 s in f32), until the summed flow + point-matching loss has
     fallen by >= 20x (measured 320x);
  2. refine the same pairs with the test loop (deepim/core/tester.py:523-598; FAST_TEST graph, hipGraph, box_rendered mask update):
     mean rotation AND translation error against the ground truth: after iteration 4 < after iteration 1 < initial;
  3. with these weights the FREE-RUNNING loops agree: ADD(hip, oracle) < 0.02 d for every pair, |se3_hip - se3_oracle| 3e-7 in the
     median and within 1e-3 for >= 95 % of the (pair, iteration) entries (see check_free_running_parity for why not "all"), next to the
     teacher-forced step check (<= 1e-3 from identical state, north_star's output bar; median 1.5e-7);
  4. the same with the training done on the bf16 matrix pipe (BASELINE configs[2]) and the test in f32.

This is the one independent piece of evidence that labels, losses, gradients, zoom inverse and pose composition are mutually
consistent: a sign or frame error in any of them trains a network that does NOT reduce the pose error."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from loop_parity import oracle_free_and_forced  # noqa: E402
from scene import make_test_config, make_train_config  # noqa: E402

N_PAIRS, B, EPOCHS, LR = 32, 16, 250, 1e-4
# epochs at LR, the rest at LR / 10.  No decay: 50 more epochs at 1e-5 take the loss from 320x to 1545x below its start, and the
# refinement of the SAME pairs gets worse (rotation 2.5 -> 4.7 deg after iteration 4): the test loop's inputs are not the training
# loop's (mask_observed is re-boxed every iteration, tester.py:579-587; training keeps the initial box), so memorising harder hurts.
DECAY_AFTER = 250
SEED = 2333


def pose_errors(poses, gt):
    R = np.einsum("bij,bkj->bik", poses[:, :, :3].astype(np.float64), gt[:, :, :3].astype(np.float64))
    c = np.clip((np.trace(R, axis1=1, axis2=2) - 1.0) / 2.0, -1.0, 1.0)
    return np.degrees(np.arccos(c)), np.linalg.norm(poses[:, :, 3].astype(np.float64) - gt[:, :, 3], axis=1)


def train_and_refine(dtype):
    from deepim.core.module import MutableModule, fit_epochs
    from deepim.core.tester import Predictor, Refiner
    from deepim.symbols.deepIM_flownet import deepIM_flownet
    from lib.pair_matching.batch_updater_py_multi import batchUpdaterPyMulti
    from lib.render_hip.render_py_multi import Render_Py
    from lib.utils import synthetic as syn

    cfg = make_train_config()
    cfg.TRAIN.optimizer = "adam"
    sym = deepIM_flownet()
    sym.get_symbol(cfg, is_train=True)
    params = sym.init_weights(cfg, {}, {}, seed=0)
    models = syn.make_models(seed=SEED, n_models=1, subdiv=4)
    K = syn.LINEMOD_K
    rm = Render_Py(None, cfg.dataset.class_name, K, meshes=models)
    batches = [syn.build_device_train_batch(rm, B, seed=SEED + 1000 * (i + 1), models=models, pixel_means=cfg.network.PIXEL_MEANS,
                                            npts=int(cfg.train_iter.NUM_3D_SAMPLE)) for i in range(N_PAIRS // B)]
    mod = MutableModule(cfg, params, B, compute_dtype=dtype)
    upd = batchUpdaterPyMulti(cfg, 480, 640, render_machine=rm)
    per_epoch = len(batches) * int(cfg.network.TRAIN_ITER_SIZE)     # optimizer updates per epoch
    hist = fit_epochs(mod, batches, upd, lambda n_update: LR if n_update <= DECAY_AFTER * per_epoch else 0.1 * LR, EPOCHS)
    trained = mod.get_params()
    del mod, upd
    torch.cuda.empty_cache()
    cfg = make_test_config(test_iter=4)
    pred = Predictor(cfg, trained, B)
    ref = Refiner(cfg, pred, rm, B, capture_graph=True)
    poses, se3 = [], []
    for b in batches:
        ref.load(b["image_observed"], b["image_rendered"], b["mask_observed"], b["mask_rendered"], b["src_pose"], b["class_index"])
        poses.append(ref.refine().cpu().numpy())
        se3.append(ref.se3_iter.cpu().numpy())
        assert int(ref.status_iter.abs().sum().item()) == 0
    poses, se3 = np.concatenate(poses, axis=1), np.concatenate(se3, axis=1)     # (4, N, 3, 4), (4, N, 7)
    gt = np.concatenate([b["pose_gt"].cpu().numpy() for b in batches])
    init = np.concatenate([b["src_pose"].cpu().numpy() for b in batches])
    err = np.array([pose_errors(init, gt)] + [pose_errors(poses[i], gt) for i in range(4)])    # (5, 2, N)
    return {"cfg": cfg, "params": trained, "models": models, "K": K, "batches": batches, "hist": hist, "poses": poses, "se3": se3,
            "init": init, "err": err}


def check_learned_and_contracts(r, tag):
    hist = r["hist"]
    first, last = hist[0, :, :2].sum(), hist[-1, :, :2].sum()
    print("{}: flow + point-matching loss per pair, first epoch {:.1f} -> last epoch {:.1f} ({:.0f}x)".format(tag, first / N_PAIRS, last / N_PAIRS, first / last))
    assert last * 20.0 <= first, (first, last)
    rot, tr = r["err"][:, 0].mean(1), r["err"][:, 1].mean(1)
    print("{}: mean rotation error (deg)  initial / after iteration 1..4: {}".format(tag, " ".join("%.2f" % v for v in rot)))
    print("{}: mean translation error (mm) initial / after iteration 1..4: {}".format(tag, " ".join("%.1f" % (1e3 * v) for v in tr)))
    assert rot[4] < rot[1] < rot[0] and tr[4] < tr[1] < tr[0], (rot, tr)
    assert rot[4] < 0.4 * rot[0] and tr[4] < 0.4 * tr[0], (rot, tr)      # 20 deg / 47 mm initial: the loop removes most of it


def check_free_running_parity(r, pairs, tag):
    """HIP loop vs oracle loop under the TRAINED weights, per pair and iteration (oracle/loop_check.loop_numbers).

    What was measured on MI355X before the bars were set (tools/learn_protocol_sweep.py, 3 schedules x 16 pairs x 3 re-renders):
    * renders of the two rasterisers from an identical pose: pixel-identical masks in 144 of 144 cases;
    * teacher-forced (same state) step error: median 1.5e-7, 4e-6 at most -- except one 6.3e-5 outlier with identical renders (a
      discrete event downstream of the render: cause not isolated; a rounded zoom-mask pixel or a LeakyReLU branch would do it);
    * free-running |se3_hip - se3_oracle|: 1e-7 .. 2e-6 for almost every (pair, iteration), and RARE jumps to 1e-4 .. 3e-3: each
      loop renders from its own pose, and a 1e-7 pose difference now and then puts one silhouette pixel on the other side -- a trained
      network answers a one-pixel change of its input with 1e-4 .. 1e-3, a CONTRACTING loop then forgets it (the random-head loops of
      tests/test_gpu_refine.py amplify it 10-200x per iteration instead).
    So: the hard bars are the north_star's 1e-3 on the outputs from identical state and ADD < 0.02 d free-running for EVERY pair;
    free-running se3 is barred by its median (1e-5) and by the share of (pair, iteration) entries within 1e-3 (>= 95 %), with 2e-2 as
    the ceiling for a one-pixel event -- "every entry within 1e-3" held in the run the test was written on (max 4.1e-4 over 32 pairs)
    but is a property of which pixels happen to flip, not of the implementation."""
    from oracle import loop_check

    cfg, models = r["cfg"], r["models"]
    pts = models[0][0].astype(np.float64)
    diam = float(np.linalg.norm(pts.max(0) - pts.min(0)))
    step, free_se3, add_same, add_free = [], [], [], []
    for i in pairs:
        b, j = r["batches"][i // B], i % B
        blobs = {k: b[k][j:j + 1].cpu().numpy() for k in ("image_observed", "image_rendered", "mask_observed", "mask_rendered", "src_pose")}
        free, forced = oracle_free_and_forced(r["params"], models[0], blobs, r["K"], cfg.network.PIXEL_MEANS, r["poses"][:, i], test_iter=4)
        n = loop_check.loop_numbers(r["init"][i], r["poses"][:, i], r["se3"][:, i], free, forced, pts, diam)
        step.append(n["step_err"]); free_se3.append(n["free_se3_err"]); add_same.append(n["add_same_state_over_d"]); add_free.append(n["add_free_over_d"])
    step, free_se3 = np.array(step), np.array(free_se3)          # (pairs, 4)
    within = float((free_se3 <= 1e-3).mean())
    print("{}: {} pairs x 4 iterations vs oracle: teacher-forced step error median {:.2e} max {:.2e}; free-running |se3 diff| median {:.2e} "
          "max {:.2e}, {:.1f} % within 1e-3; ADD / d same state max {:.2e}, free-running max {:.2e}".format(
              tag, len(step), np.median(step), step.max(), np.median(free_se3), free_se3.max(), 100 * within, max(add_same), max(add_free)))
    assert step.max() <= 1e-3 and np.median(step) <= 2e-6, (np.median(step), step.max())
    assert max(add_same) < 0.02 and max(add_free) < 0.02, (max(add_same), max(add_free))      # "ADD(-S) vs reference", every pair
    assert np.median(free_se3) <= 1e-5 and within >= 0.95 and free_se3.max() <= 2e-2, (np.median(free_se3), within, free_se3.max())
    return step, free_se3, add_free


@pytest.fixture(scope="module")
def trained_f32(hip_lib):
    assert torch.cuda.is_available()
    return train_and_refine("f32")


def test_overfit_then_the_loop_contracts(trained_f32):
    check_learned_and_contracts(trained_f32, "f32")


def test_trained_weights_free_running_loops_agree(trained_f32):
    check_free_running_parity(trained_f32, range(N_PAIRS), "f32")


def test_bf16_training_f32_test(hip_lib):
    r = train_and_refine("bf16")
    check_learned_and_contracts(r, "bf16-trained")
    check_free_running_parity(r, range(0, N_PAIRS, 2), "bf16-trained")
