"""Evaluation: pose-error primitives vs golden vectors produced by the reference's own functions, and the
aggregation of lib/dataset/LM6D_REFINE.py:329-893 vs oracle/evaluation.py (the reference's loops, restated outside the package)."""
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


def test_evaluate_pose_vs_oracle():
    """PoseEvaluator.evaluate_pose (vectorised) == oracle/evaluation.py (the loops of LM6D_REFINE.py:329-459), digit for digit"""
    from oracle import evaluation as oe

    classes, points, diam, est, gt = _scene()
    out = PoseEvaluator(classes, points, diam).evaluate_pose(_Cfg, est, gt)
    rot, tra, spc, overall, nvalid = oe.evaluate_pose(classes, est, gt, 2)
    assert out["num_valid_class"] == nvalid == 3
    np.testing.assert_array_equal(out["rot_acc"], rot)
    np.testing.assert_array_equal(out["trans_acc"], tra)
    np.testing.assert_array_equal(out["space_acc"], spc)
    for it in range(2):
        for k in ("RotAcc", "TraAcc", "SpcAcc"):
            assert out["overall"][it][k] == overall[it][k]
    assert np.all(out["rot_acc"][3] == 0)                      # class without results stays zero and is not counted
    assert out["overall"][1]["RotAcc"] > out["overall"][0]["RotAcc"]   # iteration 2 poses are closer
    eg = classes.index("eggbox")
    assert out["rot_acc"][eg, 1, 9] > 0.9                      # flipped twins were folded back
    assert 0 < rot[0, 0].min() and rot[0, 0].max() < 1         # the scene spreads over the thresholds: the table is not all 0 / 1


def test_evaluate_pose_add_and_arp2d_vs_oracle(tmp_path):
    """ADD / ADI (:461-681) and ARP-2D (:683-893) tables: fixed-threshold accuracies, per-threshold curves, Simpson areas and the
    over-classes rows against the oracle's per-pose / per-threshold loops"""
    from oracle import evaluation as oe

    classes, points, diam, est, gt = _scene(seed=9)
    ev = PoseEvaluator(classes, points, diam)
    out = ev.evaluate_pose_add(_Cfg, est, gt, output_dir=str(tmp_path))
    assert os.path.exists(os.path.join(str(tmp_path), "adi_xys.pkl"))
    per_class, overall, cc, count_all = oe.evaluate_pose_add(classes, points, diam, est, gt, 2)
    np.testing.assert_array_equal(out["count_all"], count_all)
    for k in ("0.02", "0.05", "0.10", "mean"):
        np.testing.assert_array_equal(out["count_correct"][k], cc[k])
    assert set(out["per_class"]) == set(per_class) and len(per_class) == 6
    for key, res in per_class.items():
        for k, v in res.items():
            assert out["per_class"][key][k] == v, (key, k)
    for it in range(2):
        for k, v in overall[it].items():
            assert abs(out["overall"][it][k] - v) <= 1e-12 * max(1.0, abs(v)), (it, k)
    assert out["overall"][1]["0.10"] >= out["overall"][0]["0.10"]
    assert 0 < per_class[("ape", 1)]["0.10"] < 100 and 0 < per_class[("ape", 1)]["auc"] < 100   # not a degenerate table

    arp = ev.evaluate_pose_arp_2d(_Cfg, est, gt)
    per_class, overall, cc, count_all = oe.evaluate_pose_arp_2d(classes, points, _Cfg.dataset.INTRINSIC_MATRIX, est, gt, 2)
    for k in ("2", "5", "10", "20", "mean"):
        np.testing.assert_array_equal(arp["count_correct"][k], cc[k])
    for key, res in per_class.items():
        for k, v in res.items():
            assert arp["per_class"][key][k] == v, (key, k)
    for it in range(2):
        for k, v in overall[it].items():
            assert abs(arp["overall"][it][k] - v) <= 1e-12 * max(1.0, abs(v)), (it, k)
