"""Does the system learn, and does the refinement loop contract?  (experiment behind tests/test_gpu_learns.py)

Trains from the seeded initialisation on a FIXED set of synthetic single-object pairs through fit_batch (TRAIN_ITER_SIZE inner
iterations with re-render + re-label in between: reference deepim/core/module.py:1205-1213), prints the losses per epoch and, every
--eval-every epochs, refines the same pairs with the test loop (deepim/core/tester.py:523-598) and prints the mean rotation /
translation error against the ground truth: initial, after iteration 1 .. test_iter.

    python tools/overfit_probe.py --pairs 32 --epochs 200 --opt sgd --lr 1e-4
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "mx-deepim_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)
CFG = os.path.join(PKG, "experiments", "deepim", "cfgs", "deepim_hip_LM_ape_test.yaml")


def pose_errors(poses, gt):
    """poses (B,3,4), gt (B,3,4) -> (rot err deg (B,), trans err m (B,))"""
    R = np.einsum("bij,bkj->bik", poses[:, :, :3].astype(np.float64), gt[:, :, :3].astype(np.float64))
    c = np.clip((np.trace(R, axis1=1, axis2=2) - 1.0) / 2.0, -1.0, 1.0)
    return np.degrees(np.arccos(c)), np.linalg.norm(poses[:, :, 3] - gt[:, :, 3], axis=1)


def evaluate(cfg, params, rm, batches, B, tag=""):
    import torch

    from deepim.core.tester import Predictor, Refiner

    pred = Predictor(cfg, params, B)
    ref = Refiner(cfg, pred, rm, B)
    rows = []
    for b in batches:
        ref.load(b["image_observed"], b["image_rendered"], b["mask_observed"], b["mask_rendered"], b["src_pose"], b["class_index"])
        poses = ref.refine().cpu().numpy()
        gt = b["pose_gt"].cpu().numpy()
        e = [pose_errors(b["src_pose"].cpu().numpy(), gt)] + [pose_errors(poses[i], gt) for i in range(poses.shape[0])]
        rows.append(np.array(e))          # (1 + iters, 2, B)
    e = np.concatenate(rows, axis=2)
    print("{} rot err deg  mean: {}   median: {}".format(tag, " ".join("%6.2f" % v for v in e[:, 0].mean(1)),
                                                         " ".join("%6.2f" % v for v in np.median(e[:, 0], 1))))
    print("{} trans err mm mean: {}   median: {}".format(tag, " ".join("%6.1f" % (1e3 * v) for v in e[:, 1].mean(1)),
                                                         " ".join("%6.1f" % (1e3 * v) for v in np.median(e[:, 1], 1))))
    sys.stdout.flush()
    del ref, pred
    torch.cuda.empty_cache()
    return e


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pairs", type=int, default=32)
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--epochs", type=int, default=200)
    ap.add_argument("--eval-every", type=int, default=50)
    ap.add_argument("--opt", default="sgd")
    ap.add_argument("--lr", type=float, default=1e-4)
    ap.add_argument("--momentum", type=float, default=0.975)
    ap.add_argument("--wd", type=float, default=5e-4)
    ap.add_argument("--warmup", type=int, default=0, help="updates at lr / 10 first")
    ap.add_argument("--dtype", default="f32")
    ap.add_argument("--angle-std", type=float, default=15.0)
    ap.add_argument("--xy-std", type=float, default=0.01)
    ap.add_argument("--z-std", type=float, default=0.05)
    ap.add_argument("--subdiv", type=int, default=4)
    ap.add_argument("--seed", type=int, default=2333)
    ap.add_argument("--minutes", type=float, default=0.0, help="stop training after this much wall time (0 = epochs only)")
    ap.add_argument("--save", default="")
    args = ap.parse_args()

    import torch

    from deepim.config.config import config as cfg, update_config
    from deepim.core.module import MutableModule, fit_batch
    from deepim.symbols.deepIM_flownet import deepIM_flownet
    from lib.pair_matching.batch_updater_py_multi import batchUpdaterPyMulti
    from lib.render_hip.render_py_multi import Render_Py
    from lib.utils import synthetic as syn

    update_config(CFG)
    cfg.TRAIN.optimizer, cfg.TRAIN.momentum, cfg.TRAIN.wd = args.opt, args.momentum, args.wd
    sym = deepIM_flownet()
    sym.get_symbol(cfg, is_train=True)
    params = sym.init_weights(cfg, {}, {}, seed=0)
    B = args.batch
    models = syn.make_models(seed=args.seed, n_models=1, subdiv=args.subdiv)
    K = np.asarray(cfg.dataset.INTRINSIC_MATRIX, dtype=np.float32).reshape(3, 3)
    rm = Render_Py(None, list(cfg.dataset.class_name), K, meshes=models)
    noise = dict(angle_std=args.angle_std, xy_std=args.xy_std, z_std=args.z_std)
    batches = [syn.build_device_train_batch(rm, B, seed=args.seed + 1000 * (i + 1), models=models, pixel_means=cfg.network.PIXEL_MEANS,
                                            npts=int(cfg.train_iter.NUM_3D_SAMPLE), noise=noise) for i in range(args.pairs // B)]
    mod = MutableModule(cfg, params, B, compute_dtype=args.dtype)
    upd = batchUpdaterPyMulti(cfg, 480, 640, render_machine=rm)
    evaluate(cfg, mod.get_params(), rm, batches, B, tag="epoch   0")
    t0 = time.time()
    for ep in range(1, args.epochs + 1):
        sums = np.zeros((int(cfg.network.TRAIN_ITER_SIZE), 3))
        for b in batches:
            work = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in b.items()}
            lr = args.lr * (0.1 if mod.num_update < args.warmup else 1.0)
            outs = fit_batch(mod, work, upd, lr)
            for i, o in enumerate(outs):
                sums[i] += o["loss_sums"][:3].cpu().numpy()
        n = args.pairs
        if ep % 10 == 0 or ep == 1:
            print("epoch {:4d} ({:6.1f} s) per pair: flow {}  pm {}  mask {}".format(
                ep, time.time() - t0, " ".join("%9.3f" % (v / n) for v in sums[:, 0]), " ".join("%8.4f" % (v / n) for v in sums[:, 1]),
                " ".join("%8.1f" % (v / n) for v in sums[:, 2])))
            sys.stdout.flush()
        if not np.isfinite(sums).all():
            print("diverged")
            break
        stop = args.minutes and time.time() - t0 > 60 * args.minutes
        if ep % args.eval_every == 0 or ep == args.epochs or stop:
            evaluate(cfg, mod.get_params(), rm, batches, B, tag="epoch {:3d}".format(ep))
        if stop:
            break
    if args.save:
        np.savez(args.save, **mod.get_params())


if __name__ == "__main__":
    main()
