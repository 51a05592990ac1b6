"""Sweep of training protocols for tests/test_gpu_learns.py: which fixed-set schedule gives a network that contracts AND is smooth enough
for the free-running HIP / oracle loops to stay within 1e-3 on se3?  Runs the test's own functions with other constants.
    python tools/learn_protocol_sweep.py "pairs,epochs,decay_after,lr,dtype" ..."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "mx-deepim_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)

import numpy as np  # noqa: E402
import test_gpu_learns as T  # noqa: E402

for spec in sys.argv[1:]:
    pairs, epochs, decay, lr, dtype = spec.split(",")
    T.N_PAIRS, T.EPOCHS, T.DECAY_AFTER, T.LR = int(pairs), int(epochs), int(decay), float(lr)
    tag = "[{}]".format(spec)
    r = T.train_and_refine(dtype)
    try:
        T.check_learned_and_contracts(r, tag)
    except AssertionError as e:
        print(tag, "contract check FAILED:", str(e)[:200])
    sample = list(range(0, T.N_PAIRS, max(1, T.N_PAIRS // 16)))
    # per (pair, iteration): teacher-forced step error, and how the two renders of that iteration's input differ
    import torch
    from oracle import loop_check, native
    from lib.render_hip.render_py_multi import Render_Py

    cfg, models = r["cfg"], r["models"]
    pts = models[0][0].astype(np.float64)
    diam = float(np.linalg.norm(pts.max(0) - pts.min(0)))
    rm = Render_Py(None, cfg.dataset.class_name, r["K"], meshes=models)
    rows = []
    for i in sample:
        b, j = r["batches"][i // T.B], i % T.B
        blobs = {k: b[k][j:j + 1].cpu().numpy() for k in ("image_observed", "image_rendered", "mask_observed", "mask_rendered", "src_pose")}
        free, forced = loop_check.oracle_free_and_forced(r["params"], models[0], blobs, r["K"], cfg.network.PIXEL_MEANS, r["poses"][:, i], test_iter=4)
        n = loop_check.loop_numbers(r["init"][i], r["poses"][:, i], r["se3"][:, i], free, forced, pts, diam)
        for it in range(1, 4):   # iteration `it` starts from the render at poses[it - 1]
            pose = r["poses"][it - 1, i]
            mask = torch.empty((1, 1, 480, 640), device="cuda:0")
            bbox = torch.empty((1, 4), dtype=torch.int32, device="cuda:0")
            rm.render_batch(torch.zeros(1, dtype=torch.int32, device="cuda:0"), torch.from_numpy(pose[None].astype(np.float32)).cuda(), mask=mask, bbox=bbox)
            _, d = native.render(*models[0], pose[:, :3], pose[:, 3], r["K"])
            mo = (d > 0.2)
            mh = mask[0, 0].cpu().numpy() > 0
            ys, xs = np.nonzero(mo)
            ob = [xs.min(), ys.min(), xs.max(), ys.max()]
            rows.append((i, it, n["step_err"][it], int((mo != mh).sum()), bbox[0].cpu().numpy().tolist(), ob, n["free_se3_err"][it]))
    rows.sort(key=lambda x: -x[2])
    print(tag, "teacher-forced step errors, worst first: (pair, iter, step err, differing mask pixels, hip bbox, oracle bbox, free se3 err)")
    for x in rows[:8]:
        print(tag, "   ", x)
    flips = [x for x in rows if x[3] > 0]
    print(tag, "{} of {} renders differ in >= 1 mask pixel; median step err with / without a flip: {:.2e} / {:.2e}; max free se3 err {:.2e}, max ADD free {:.2e}".format(
        len(flips), len(rows), np.median([x[2] for x in flips]) if flips else 0.0, np.median([x[2] for x in rows if x[3] == 0]),
        max(x[6] for x in rows), 0.0))
    sys.stdout.flush()
