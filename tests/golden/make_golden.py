"""Generate golden vectors by IMPORTING the reference's own numpy modules (build container only).

Run:  python tests/golden/make_golden.py        (needs /root/reference; never runs on the GPU box)
Writes small .npz fixtures next to this file.  The reference's source never leaves the container;
only inputs and expected outputs are stored.

Modules imported from /root/reference:
  lib/pair_matching/RT_transform.py  (numpy-2 alias shim for its module-level np.float uses, :246-247)
  lib/pair_matching/flow.py (calc_flow), lib/utils/pose_error.py (add, adi, arp_2d, re, te),
  lib/utils/get_min_rect.py, lib/utils/projection.py (se3_mul, se3_inverse, backproject_camera), lib/utils/mask_dilate.py,
  deepim/core/callback.py (Speedometer)
One function is EXECUTED without importing its module: calc_EPE_one_pair of deepim/core/tester.py:719-736 (pure numpy; the module
imports mxnet / cv2 / glumpy at its top and cannot load here) -- epe_vectors() takes that one FunctionDef out of the file with `ast`,
compiles it in memory and calls it; nothing of it is written anywhere.
"""
import os
import sys

import numpy as np

np.float = float  # noqa: shim for RT_transform.py:246-247 under numpy 2
np.int = int
np.maximum_sctype = lambda t: np.longdouble
sys.path.insert(0, "/root/reference")

from lib.pair_matching import RT_transform as RT  # noqa: E402
from lib.pair_matching.flow import calc_flow  # noqa: E402
from lib.utils.pose_error import add, adi, arp_2d, re, te  # noqa: E402
from lib.utils.get_min_rect import get_min_rect  # noqa: E402
from lib.utils.projection import backproject_camera, se3_inverse, se3_mul  # noqa: E402
from lib.utils.mask_dilate import mask_dilate  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
K = np.array([[572.4114, 0, 325.2611], [0, 573.57043, 242.04899], [0, 0, 1]])


def rand_pose(rng):
    q = rng.normal(size=4)
    R = RT.quat2mat(q / np.linalg.norm(q))
    t = np.array([rng.uniform(-0.2, 0.2), rng.uniform(-0.15, 0.15), rng.uniform(0.5, 1.3)])
    return np.concatenate([R, t[:, None]], axis=1)


def se3_vectors():
    rng = np.random.default_rng(2333)
    n = 64
    out = {k: [] for k in ("pose_src", "pose_tgt", "quat_raw", "trans_delta")}
    coords = ["MODEL", "CAMERA", "CAMERA_NEW", "NAIVE"]
    res = {c: {"compose": [], "delta_q": [], "delta_t": [], "delta_R": []} for c in coords}
    T_means = np.array([0.0, 0.0, 0.0])
    T_stds = np.array([1.0, 1.0, 1.0])
    T_means2 = np.array([0.01, -0.02, 0.03])
    T_stds2 = np.array([0.5, 2.0, 1.5])
    res2 = {"compose": [], "delta_q": [], "delta_t": []}
    dist = []
    for i in range(n):
        ps, pt = rand_pose(rng), rand_pose(rng)
        q = rng.normal(size=4) * rng.uniform(0.2, 3.0)  # un-normalised, as the test graph emits it
        td = rng.normal(size=3) * np.array([0.05, 0.05, 0.2])
        out["pose_src"].append(ps); out["pose_tgt"].append(pt); out["quat_raw"].append(q); out["trans_delta"].append(td)
        for c in coords:
            res[c]["compose"].append(RT.RT_transform(ps, q, td, T_means, T_stds, c))
            r, t = RT.calc_RT_delta(ps, pt, T_means, T_stds, c, "QUAT")
            res[c]["delta_q"].append(r); res[c]["delta_t"].append(t)
            rm, _ = RT.calc_RT_delta(ps, pt, T_means, T_stds, c, "MATRIX")
            res[c]["delta_R"].append(rm)
        res2["compose"].append(RT.RT_transform(ps, q, td, T_means2, T_stds2, "CAMERA"))
        r, t = RT.calc_RT_delta(ps, pt, T_means2, T_stds2, "CAMERA", "QUAT")
        res2["delta_q"].append(r); res2["delta_t"].append(t)
        dist.append(RT.calc_rt_dist_m(ps, pt))
    save = {k: np.array(v) for k, v in out.items()}
    for c in coords:
        for k, v in res[c].items():
            save["{}_{}".format(c, k)] = np.array(v)
    for k, v in res2.items():
        save["ms_CAMERA_{}".format(k)] = np.array(v)
    save["T_means2"], save["T_stds2"] = T_means2, T_stds2
    save["rt_dist"] = np.array(dist)
    # analytic constants from the docstrings (RT_transform.py:413-418, :545-547)
    save["quat2mat_id"] = RT.quat2mat([1, 0, 0, 0])
    save["quat2mat_x180"] = RT.quat2mat([0, 1, 0, 0])
    save["euler2quat_123_ryxz"] = RT.euler2quat(1, 2, 3, "ryxz")
    # mat2quat on exact and slightly noisy rotations (sign rule w >= 0)
    Rs = np.array([rand_pose(rng)[:, :3] for _ in range(32)])
    save["m2q_R"] = Rs
    save["m2q_q"] = np.array([RT.mat2quat(R) for R in Rs])
    # se3_mul / se3_inverse (float32 outputs)
    save["se3_mul"] = np.array([se3_mul(a, b) for a, b in zip(save["pose_src"], save["pose_tgt"])])
    save["se3_inv"] = np.array([se3_inverse(a) for a in save["pose_src"]])
    np.savez_compressed(os.path.join(HERE, "se3_golden.npz"), **save)


def se3_euler_vectors():
    """EULER deltas (RT_transform.py:39-40, :139-140): calc_RT_delta(..., "EULER") and RT_transform with a 3-number rotation, on the
    poses of se3_golden.npz plus exact gimbal-lock cases (aj = +-pi/2) for mat2euler's second branch."""
    g = np.load(os.path.join(HERE, "se3_golden.npz"))
    rng = np.random.default_rng(99)
    coords = ["MODEL", "CAMERA", "CAMERA_NEW", "NAIVE"]
    z3, o3 = np.zeros(3), np.ones(3)
    n = g["pose_src"].shape[0]
    eul = rng.uniform(-np.pi, np.pi, size=(n, 3)) * np.array([1.0, 0.5, 1.0])
    save = {"euler": eul}
    for c in coords:
        save[c + "_compose"] = np.array([RT.RT_transform(g["pose_src"][i], eul[i], g["trans_delta"][i], z3, o3, c) for i in range(n)])
        # calc_RT_delta(..., "EULER") = mat2euler of the MATRIX residual (:39-40).  Under numpy 2 the reference's mat2euler raises
        # whenever np.array(mat, float64, copy=False) would have to copy (:357; numpy 1 copied silently) -- the float32 residual of
        # NAIVE always, views sometimes -- so the residual is taken in MATRIX mode and handed over as the float64 array numpy 1
        # would have made of it.
        rt = [RT.calc_RT_delta(g["pose_src"][i], g["pose_tgt"][i], z3, o3, c, "MATRIX") for i in range(n)]
        save[c + "_delta_e"] = np.array([np.array(RT.mat2euler(np.ascontiguousarray(r, dtype=np.float64))) for r, _ in rt])
        save[c + "_delta_t"] = np.array([t for _, t in rt])
    lock = np.array([[0.3, np.pi / 2, -0.7], [1.1, -np.pi / 2, 0.4], [0.0, np.pi / 2, 0.0]])
    Rl = np.array([RT.euler2mat(*e) for e in lock])
    save["lock_R"] = Rl
    save["lock_e"] = np.array([np.array(RT.mat2euler(R)) for R in Rl])
    save["e2m_R"] = np.array([RT.euler2mat(*e) for e in eul])
    np.savez_compressed(os.path.join(HERE, "se3_euler_golden.npz"), **save)


def synth_depth(rng, pose, H=120, W=160, Ks=None):
    """depth of a sphere of radius .08 m at pose translation (analytic ray cast), small image for fixture size"""
    c = pose[:, 3]
    ys, xs = np.meshgrid(np.arange(H), np.arange(W), indexing="ij")
    d = np.stack([(xs - Ks[0, 2]) / Ks[0, 0], (ys - Ks[1, 2]) / Ks[1, 1], np.ones_like(xs, dtype=np.float64)], axis=-1)
    a = (d * d).sum(-1)
    b = -2 * (d @ c)
    cc = c @ c - 0.08 ** 2
    disc = b * b - 4 * a * cc
    t = np.where(disc > 0, (-b - np.sqrt(np.maximum(disc, 0))) / (2 * a), 0.0)
    return (t * (disc > 0)).astype(np.float32)  # z-depth since d_z = 1


def flow_vectors():
    rng = np.random.default_rng(7)
    H, W = 120, 160
    Ks = K.copy()
    Ks[:2] *= 0.25
    save = {"K": Ks}
    ds, dt, ps, pt, fl, vis = [], [], [], [], [], []
    for i in range(6):
        p0 = rand_pose(rng)
        p0[:, 3] = [rng.uniform(-0.05, 0.05), rng.uniform(-0.04, 0.04), rng.uniform(0.5, 0.9)]
        p1 = p0.copy()
        dq = np.array([1.0, 0, 0, 0]) + rng.normal(size=4) * 0.05
        p1[:, :3] = RT.quat2mat(dq / np.linalg.norm(dq)) @ p0[:, :3]
        p1[:, 3] += rng.normal(size=3) * np.array([0.01, 0.01, 0.03])
        d0, d1 = synth_depth(rng, p0, H, W, Ks), synth_depth(rng, p1, H, W, Ks)
        f, v, _ = calc_flow(d0, p0, p1, Ks, d1, thresh=3e-3, standard_rep=False)
        ds.append(d0); dt.append(d1); ps.append(p0); pt.append(p1); fl.append(f); vis.append(v)
    save.update(depth_src=np.array(ds), depth_tgt=np.array(dt), pose_src=np.array(ps), pose_tgt=np.array(pt),
                flow=np.array(fl, dtype=np.float32), visible=np.array(vis, dtype=np.float32))
    np.savez_compressed(os.path.join(HERE, "flow_golden.npz"), **save)


def pose_error_vectors():
    rng = np.random.default_rng(11)
    pts = rng.normal(size=(500, 3)) * 0.05
    pe, pg, a, s, r2, rr, tt = [], [], [], [], [], [], []
    for i in range(16):
        g = rand_pose(rng)
        e = g.copy()
        dq = np.array([1.0, 0, 0, 0]) + rng.normal(size=4) * 0.03
        e[:, :3] = RT.quat2mat(dq / np.linalg.norm(dq)) @ g[:, :3]
        e[:, 3] += rng.normal(size=3) * 0.01
        pe.append(e); pg.append(g)
        a.append(add(e[:, :3], e[:, 3], g[:, :3], g[:, 3], pts))
        s.append(adi(e[:, :3], e[:, 3], g[:, :3], g[:, 3], pts))
        r2.append(arp_2d(e[:, :3], e[:, 3], g[:, :3], g[:, 3], pts, K))
        rr.append(re(e[:, :3], g[:, :3]))
        tt.append(te(e[:, 3], g[:, 3]))
    np.savez_compressed(os.path.join(HERE, "pose_error_golden.npz"), pts=pts, pose_est=np.array(pe), pose_gt=np.array(pg),
                        add=np.array(a), adi=np.array(s), arp_2d=np.array(r2), re=np.array(rr), te=np.array(tt), K=K)


def min_rect_vectors():
    rng = np.random.default_rng(5)
    masks, rects = [], []
    for i in range(12):
        m = np.zeros((48, 64), dtype=np.float32)
        y0, x0 = rng.integers(0, 40), rng.integers(0, 56)
        h, w = rng.integers(1, 48 - y0 + 1), rng.integers(1, 64 - x0 + 1)
        blob = (rng.uniform(size=(h, w)) > 0.6).astype(np.float32)
        if blob.sum() == 0:
            blob[0, 0] = 1
        m[y0:y0 + h, x0:x0 + w] = blob
        masks.append(m)
        rects.append(get_min_rect(m))
    np.savez_compressed(os.path.join(HERE, "min_rect_golden.npz"), masks=np.array(masks), rects=np.array(rects))


def data_layer_vectors():
    """N4 (lib/utils/image.py helpers that import here): mask_dilate under a seeded numpy global RNG -- the seeds cover every value of
    `direction` --, with both thickness bounds the callers use (10: image.py:346, :488); backproject_camera; calc_flow's X_valid."""
    rng = np.random.default_rng(21)
    masks = []
    for i in range(3):
        m = np.zeros((60, 80), dtype=np.float64)
        y0, x0 = rng.integers(12, 25), rng.integers(12, 35)
        m[y0:y0 + rng.integers(8, 25), x0:x0 + rng.integers(8, 30)] = 1.0
        m[y0 + 3:y0 + 6, x0 + 2:x0 + 5] = 0.0  # a hole: the dilation rules look at != 0 / == 0 transitions
        masks.append(m)
    seeds, outs, which, thick = [], [], [], []
    for seed in list(range(16)) + [17, 41]:   # first draws 0..9 all occur (seed 41 gives direction 0)
        for k, m in enumerate(masks):
            for mt in (10, 4):
                np.random.seed(seed)
                outs.append(mask_dilate(m, max_thickness=mt))
                seeds.append(seed); which.append(k); thick.append(mt)
    Ks = K.copy()
    Ks[:2] *= 0.125
    depth = (rng.uniform(0.4, 1.2, size=(60, 80)) * (rng.uniform(size=(60, 80)) > 0.3)).astype(np.float32)
    X = backproject_camera(depth, Ks)
    p0, p1 = rand_pose(rng), None
    p0[:, 3] = [0.01, -0.02, 0.7]
    p1 = p0.copy()
    p1[:, 3] += [0.004, 0.002, 0.01]
    d0, d1 = synth_depth(rng, p0, 60, 80, Ks), synth_depth(rng, p1, 60, 80, Ks)
    f_std, v_std, X_valid = calc_flow(d0, p0, p1, Ks, d1, thresh=3e-3, standard_rep=True)
    np.savez_compressed(os.path.join(HERE, "data_golden.npz"), masks=np.array(masks), dil_seed=np.array(seeds), dil_mask=np.array(which),
                        dil_thick=np.array(thick), dil_out=np.array(outs), K=Ks, depth=depth, backproject=X.astype(np.float64),
                        cf_depth_src=d0, cf_depth_tgt=d1, cf_pose_src=p0, cf_pose_tgt=p1, cf_flow_std=f_std.astype(np.float32),
                        cf_visible=v_std.astype(np.float32), cf_X_valid=np.asarray(X_valid, dtype=np.float64))


def callback_vectors():
    """N2 (deepim/core/callback.py, importable: time + logging only): the lines Speedometer prints for a scripted sequence of batch-end
    callbacks under a scripted clock (the module's `time.time` is replaced for the run) -- with a metric, without one, across an epoch
    boundary (batch counter going backwards) and with a counter that skips multiples of `frequent`."""
    import json
    from collections import namedtuple

    from deepim.core import callback as ref_cb

    Param = namedtuple("Param", ["epoch", "nbatch", "eval_metric", "locals"])

    class Metric(object):
        def __init__(self, names, values):
            self.names, self.values = names, values

        def get(self):
            return self.names, self.values

    scripts = []
    rng = np.random.default_rng(7)
    for case, (batch, freq, with_metric) in enumerate([(16, 3, True), (4, 5, False), (64, 2, True), (1, 1, True)]):
        events, t, clock = [], 100.0 * (case + 1), []
        for epoch in range(3):
            n = int(rng.integers(4, 12))
            counts = list(range(n)) if case != 2 else [int(c) for c in np.cumsum(rng.integers(1, 3, size=n))]
            for c in counts:
                vals = [float(v) for v in rng.uniform(0.0, 3.0, size=3)]
                events.append({"epoch": epoch, "nbatch": int(c), "metric": vals if with_metric else None})
        names = ["Flow_L2Loss", "MaskLoss", "PointMatchingLoss"]
        ticks = [float(v) for v in np.cumsum(rng.uniform(0.05, 0.4, size=3 * len(events) + 8)) + t]
        it = iter(ticks)
        lines = []
        real_time, real_print = ref_cb.time.time, None
        ref_cb.time.time = lambda: next(it)
        import builtins
        real_print = builtins.print
        builtins.print = lambda *a, **k: lines.append(" ".join(str(x) for x in a))
        try:
            sp = ref_cb.Speedometer(batch, freq)
            per_event = []
            for e in events:
                before = len(lines)
                sp(Param(e["epoch"], e["nbatch"], Metric(names, e["metric"]) if e["metric"] is not None else None, None))
                per_event.append(lines[before] if len(lines) > before else None)
        finally:
            ref_cb.time.time = real_time
            builtins.print = real_print
        scripts.append({"batch_size": batch, "frequent": freq, "metric_names": names, "events": events, "clock": ticks, "lines": per_event})
    with open(os.path.join(HERE, "callback_golden.json"), "w") as f:
        json.dump(scripts, f, indent=0)


def epe_vectors():
    """Test-time flow error (tester.py:500-512): the reference's own calc_EPE_one_pair (:719-736) on the [flow, visible, bg] list that
    par_generate_gt (:706-716) builds from the reference's calc_flow, with predictions stored as float16 like tester.py:485-487.
    Inputs = the depth pairs / poses of flow_golden.npz; the un-rounded float32 predictions are saved so that the code under test does
    its own float16 rounding."""
    import ast

    src = open("/root/reference/deepim/core/tester.py").read()
    fn = [n for n in ast.parse(src).body if isinstance(n, ast.FunctionDef) and n.name == "calc_EPE_one_pair"]
    assert len(fn) == 1
    ns = {"np": np}
    exec(compile(ast.Module(body=fn, type_ignores=[]), "tester.py:calc_EPE_one_pair", "exec"), ns)
    calc_EPE_one_pair = ns["calc_EPE_one_pair"]
    g = np.load(os.path.join(HERE, "flow_golden.npz"))
    rng = np.random.default_rng(29)
    preds, outs = [], []
    for i in range(g["depth_src"].shape[0]):
        d0, d1 = g["depth_src"][i], g["depth_tgt"][i]
        flow, visible, _ = calc_flow(d0, g["pose_src"][i], g["pose_tgt"][i], g["K"], d1, thresh=3e-3, standard_rep=False)
        flow_list = [flow, visible, np.logical_and(visible == 0, d0 == 0)]          # tester.py:712-716
        pred = (flow + rng.normal(size=flow.shape) * rng.uniform(0.05, 3.0) + (rng.uniform(size=flow.shape) < 0.02) * 40.0).astype(np.float32)
        cur = {"flow": pred.astype("float16")}                                        # tester.py:485-487
        r = calc_EPE_one_pair(cur, {"flow": flow_list}, "flow")
        preds.append(pred)
        outs.append([r["epe_all"], r["num_all"], r["epe_viz"], r["num_viz"], r["epe_vizbg"], r["num_vizbg"]])
    np.savez_compressed(os.path.join(HERE, "epe_golden.npz"), pred=np.array(preds), out=np.array(outs, dtype=np.float64))


if __name__ == "__main__":
    if "--epe-only" in sys.argv:
        epe_vectors()
        sys.exit(0)
    if "--data-only" in sys.argv:
        data_layer_vectors()
        sys.exit(0)
    if "--euler-only" in sys.argv:
        se3_euler_vectors()
        sys.exit(0)
    if "--callback-only" in sys.argv:
        callback_vectors()
        sys.exit(0)
    se3_vectors()
    se3_euler_vectors()
    flow_vectors()
    pose_error_vectors()
    min_rect_vectors()
    data_layer_vectors()
    callback_vectors()
    epe_vectors()
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(HERE, f)))
