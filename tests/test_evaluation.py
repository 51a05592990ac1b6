"""Evaluation: pose-error primitives vs golden vectors produced by the reference's own functions, and the
aggregation of lib/dataset/LM6D_REFINE.py:329-830 vs a literal loop restatement of it on seeded poses."""
import os

import numpy as np

from lib.dataset.evaluation import PoseEvaluator, RT_Z, se3_mul
from lib.utils import pose_error as pe

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_primitives_vs_reference_golden():
    g = np.load(os.path.join(G, "pose_error_golden.npz"))
    s = np.load(os.path.join(G, "se3_golden.npz"))
    for i in range(len(g["pose_est"])):
        e, t = g["pose_est"][i], g["pose_gt"][i]
        np.testing.assert_allclose(pe.add(e[:, :3], e[:, 3], t[:, :3], t[:, 3], g["pts"]), g["add"][i], rtol=1e-12)
        np.testing.assert_allclose(pe.adi(e[:, :3], e[:, 3], t[:, :3], t[:, 3], g["pts"]), g["adi"][i], rtol=1e-12)
        np.testing.assert_allclose(pe.arp_2d(e[:, :3], e[:, 3], t[:, :3], t[:, 3], g["pts"], g["K"]), g["arp_2d"][i], rtol=1e-12)
        np.testing.assert_allclose(pe.re(e[:, :3], t[:, :3]), g["re"][i], rtol=1e-6, atol=1e-6)   # logm vs closed form
        np.testing.assert_allclose(pe.te(e[:, 3], t[:, 3]), g["te"][i], rtol=1e-12)
    for i in range(len(s["pose_src"])):
        rd, td = pe.calc_rt_dist_m(s["pose_src"][i], s["pose_tgt"][i])
        np.testing.assert_allclose([rd, td], s["rt_dist"][i], rtol=1e-6, atol=1e-6)
        np.testing.assert_allclose(se3_mul(s["pose_src"][i], s["pose_tgt"][i]), s["se3_mul"][i], atol=1e-6)


class _Cfg(object):
    class TEST(object):
        test_iter = 2

    class dataset(object):
        INTRINSIC_MATRIX = np.array([[572.4114, 0, 325.2611], [0, 573.57043, 242.04899], [0, 0, 1]])


def _rand_R(rng, max_deg):
    ax = rng.normal(size=3)
    ax /= np.linalg.norm(ax)
    a = np.radians(rng.uniform(0, max_deg))
    Kx = np.array([[0, -ax[2], ax[1]], [ax[2], 0, -ax[0]], [-ax[1], ax[0], 0]])
    return np.eye(3) + np.sin(a) * Kx + (1 - np.cos(a)) * Kx @ Kx


def _scene(seed=5):
    rng = np.random.default_rng(seed)
    classes = ["ape", "eggbox", "glue", "cat"]                 # add, adi + z-flip, adi, (no results)
    points = {c: rng.normal(size=(200, 3)) * 0.04 for c in classes}
    diam = {c: 0.1 + 0.02 * i for i, c in enumerate(classes)}
    est = [[[] for _ in range(2)] for _ in classes]
    gt = [[[] for _ in range(1)] for _ in classes]
    for ci, c in enumerate(classes[:3]):
        for j in range(25):
            R = _rand_R(rng, 180)
            t = np.array([rng.uniform(-.2, .2), rng.uniform(-.2, .2), rng.uniform(.6, 1.2)])
            g = np.concatenate([R, t[:, None]], 1)
            gt[ci][0].append(g)
            for it, (deg, tr) in enumerate(((12, 0.03), (4, 0.008))):
                e = np.concatenate([_rand_R(rng, deg) @ R, (t + rng.normal(size=3) * tr)[:, None]], 1)
                if c == "eggbox" and j % 3 == 0:
                    e = se3_mul(e, RT_Z)                       # the symmetric twin: must be scored as if un-flipped
                est[ci][it].append(e)
    return classes, points, diam, est, gt


def test_evaluate_pose_vs_loop_restatement():
    classes, points, diam, est, gt = _scene()
    out = PoseEvaluator(classes, points, diam).evaluate_pose(_Cfg, est, gt)
    assert out["num_valid_class"] == 3
    rot_t, tr_t = np.arange(1, 11, 1), np.arange(0.01, 0.11, 0.01)
    for ci, c in enumerate(classes[:3]):
        for it in range(2):
            rd, td = [], []
            for e, g in zip(est[ci][it], gt[ci][0]):
                r, t = pe.calc_rt_dist_m(e, g)
                if c == "eggbox" and r > 90:
                    r, t = pe.calc_rt_dist_m(se3_mul(e, RT_Z), g)
                rd.append(r); td.append(t)
            rd, td = np.array(rd)[:, None], np.array(td)[:, None]
            for k in range(10):
                assert out["rot_acc"][ci, it, k] == np.mean(rd < rot_t[k])
                assert out["trans_acc"][ci, it, k] == np.mean(td < tr_t[k])
                assert out["space_acc"][ci, it, k] == np.mean(np.logical_and(rd < rot_t[k], td < tr_t[k]))
    assert np.all(out["rot_acc"][3] == 0)                      # class without results stays zero and is not counted
    assert out["overall"][1]["RotAcc"] > out["overall"][0]["RotAcc"]   # iteration 2 poses are closer
    eg = classes.index("eggbox")
    assert out["rot_acc"][eg, 1, 9] > 0.9                      # flipped twins were folded back


def test_evaluate_pose_add_and_arp2d_vs_loop_restatement(tmp_path):
    classes, points, diam, est, gt = _scene(seed=9)
    ev = PoseEvaluator(classes, points, diam)
    out = ev.evaluate_pose_add(_Cfg, est, gt, output_dir=str(tmp_path))
    assert os.path.exists(os.path.join(str(tmp_path), "adi_xys.pkl"))
    dx = 0.0001
    base = np.arange(0, 0.1, dx).astype(np.float32)
    for ci, c in enumerate(classes[:3]):
        for it in range(2):
            cnt = {"0.02": 0, "0.05": 0, "0.10": 0}
            curve = np.zeros(len(base), np.float32)
            thr = base * np.float32(diam[c])
            for e, g in zip(est[ci][it], gt[ci][0]):
                fn = pe.adi if c in ("eggbox", "glue") else pe.add
                err = fn(e[:3, :3], e[:, 3], g[:3, :3], g[:, 3], points[c])
                for k, f in (("0.02", 0.02), ("0.05", 0.05), ("0.10", 0.10)):
                    cnt[k] += err < np.float32(f * diam[c])
                for ti in range(len(thr)):                     # the reference's per-threshold loop (:538-540)
                    if err < thr[ti]:
                        curve[ti] += 1
            res = out["per_class"][(c, it)]
            for k in cnt:
                assert res[k] == 100.0 * cnt[k] / 25
            np.testing.assert_array_equal(out["count_correct"]["mean"][ci, it], curve)
            assert 0.0 <= res["auc"] <= 100.0
    assert out["overall"][1]["0.10"] >= out["overall"][0]["0.10"]
    arp = ev.evaluate_pose_arp_2d(_Cfg, est, gt)
    K = _Cfg.dataset.INTRINSIC_MATRIX
    for ci, c in enumerate(classes[:3]):
        errs = []
        for e, g in zip(est[ci][1], gt[ci][0]):
            if c == "eggbox" and pe.re(e[:3, :3], g[:3, :3]) > 90:
                e = se3_mul(e, RT_Z)
            errs.append(pe.arp_2d(e[:3, :3], e[:, 3], g[:3, :3], g[:, 3], points[c], K))
        errs = np.array(errs)
        for k in ("2", "5", "10", "20"):
            assert arp["per_class"][(c, 1)][k] == 100.0 * np.sum(errs < float(k)) / 25
