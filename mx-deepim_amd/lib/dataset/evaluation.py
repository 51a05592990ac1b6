"""Pose evaluation of the refinement results, as lib/dataset/LM6D_REFINE.py:329-830 computes it for the README tables:
rotation / translation / joint accuracies at (k deg, k cm), ADD(-S) at 0.02 / 0.05 / 0.10 of the diameter plus the AUC over
[0, 0.1 d] (Simpson), and the 2-D re-projection accuracy at 2 / 5 / 10 / 20 px plus its AUC over [0, 50 px].

Inputs have the reference's layout: all_poses_est[cls][iter] and all_poses_gt[cls][0] are lists of 3x4 poses.  Every method
returns its numbers as a dict and logs the reference's lines through `logger` / print.  Thresholding is vectorised (the
reference loops over 1000 thresholds per pose); pose errors come from lib/utils/pose_error.py."""
from __future__ import print_function, division

import os
import pickle

import numpy as np

from lib.utils.pose_error import add, adi, arp_2d, calc_rt_dist_m, re

try:
    from scipy.integrate import simpson as _simps
except ImportError:  # older scipy
    from scipy.integrate import simps as _simps

SYM_CLASSES = ("eggbox", "glue", "bowl", "cup")            # ADI instead of ADD (:516)
RT_Z = np.array([[-1, 0, 0, 0], [0, -1, 0, 0], [0, 0, 1, 0]], dtype=np.float64)  # eggbox: 180 deg about z (:357-362)


def se3_mul(RT1, RT2):
    """lib/utils/projection.py: [R1 R2 | R1 t2 + t1]"""
    R = RT1[:, :3].dot(RT2[:, :3])
    t = RT1[:, :3].dot(RT2[:, 3]) + RT1[:, 3]
    return np.concatenate([R, t.reshape(3, 1)], axis=1)


def print_and_log(s, logger=None):
    print(s)
    if logger:
        logger.info(s)


class PoseEvaluator(object):
    def __init__(self, classes, points, diameters):
        """points: dict class -> (N,3) model points; diameters: dict class -> metres (LM6D_REFINE._points / ._diameters)"""
        self.classes = list(classes)
        self.num_classes = len(self.classes)
        self._points = points
        self._diameters = diameters

    def _valid(self, all_poses_est, all_poses_gt, cls_idx):
        return bool(len(all_poses_est[cls_idx][0]) and len(all_poses_gt[cls_idx][0]))

    # ------------------------------------------------------------------------------------------------ :329-459
    def evaluate_pose(self, config, all_poses_est, all_poses_gt, logger=None):
        print_and_log("evaluating pose", logger)
        rot_thresh_list = np.arange(1, 11, 1)
        trans_thresh_list = np.arange(0.01, 0.11, 0.01)
        num_metric = len(rot_thresh_list)
        num_iter = config.TEST.test_iter
        rot_acc = np.zeros((self.num_classes, num_iter, num_metric))
        trans_acc = np.zeros((self.num_classes, num_iter, num_metric))
        space_acc = np.zeros((self.num_classes, num_iter, num_metric))
        num_valid_class = 0
        show_list = [1, 4, 9]
        for cls_idx, cls_name in enumerate(self.classes):
            if not self._valid(all_poses_est, all_poses_gt, cls_idx):
                continue
            num_valid_class += 1
            gts = all_poses_gt[cls_idx][0]
            for iter_i in range(num_iter):
                ests = all_poses_est[cls_idx][iter_i]
                rd, td = np.zeros(len(gts)), np.zeros(len(gts))
                for j in range(len(gts)):
                    r, t = calc_rt_dist_m(ests[j], gts[j])
                    if cls_name == "eggbox" and r > 90:
                        r, t = calc_rt_dist_m(se3_mul(ests[j], RT_Z), gts[j])
                    rd[j], td[j] = r, t
                r_ok = rd[:, None] < rot_thresh_list[None, :]
                t_ok = td[:, None] < trans_thresh_list[None, :]
                rot_acc[cls_idx, iter_i] = r_ok.mean(0)
                trans_acc[cls_idx, iter_i] = t_ok.mean(0)
                space_acc[cls_idx, iter_i] = np.logical_and(r_ok, t_ok).mean(0)
            print_and_log("------------ {} -----------".format(cls_name), logger)
            print_and_log("{:>24}: {:>7}, {:>7}, {:>7}".format("[rot_thresh, trans_thresh", "RotAcc", "TraAcc", "SpcAcc"), logger)
            for iter_i in range(num_iter):
                print_and_log("** iter {} **".format(iter_i + 1), logger)
                print_and_log("{:<16}{:>8}: {:>7.2f}, {:>7.2f}, {:>7.2f}".format(
                    "average_accuracy", "[{:>2}, {:>5.2f}]".format(-1, -1), np.mean(rot_acc[cls_idx, iter_i, :]) * 100,
                    np.mean(trans_acc[cls_idx, iter_i, :]) * 100, np.mean(space_acc[cls_idx, iter_i, :]) * 100), logger)
                for show_idx in show_list:
                    print_and_log("{:>16}{:>8}: {:>7.2f}, {:>7.2f}, {:>7.2f}".format(
                        "average_accuracy", "[{:>2}, {:>5.2f}]".format(rot_thresh_list[show_idx], trans_thresh_list[show_idx]),
                        rot_acc[cls_idx, iter_i, show_idx] * 100, trans_acc[cls_idx, iter_i, show_idx] * 100,
                        space_acc[cls_idx, iter_i, show_idx] * 100), logger)
        overall = []
        for iter_i in range(num_iter):
            n = max(num_valid_class, 1)
            row = {"RotAcc": np.sum(rot_acc[:, iter_i, :]) / (n * num_metric) * 100,
                   "TraAcc": np.sum(trans_acc[:, iter_i, :]) / (n * num_metric) * 100,
                   "SpcAcc": np.sum(space_acc[:, iter_i, :]) / (n * num_metric) * 100}
            overall.append(row)
            print_and_log("---------- performance over {} classes -----------".format(num_valid_class), logger)
            print_and_log("** iter {} **".format(iter_i + 1), logger)
            print_and_log("{:<16}{:>8}: {:>7.2f}, {:>7.2f}, {:>7.2f}".format(
                "average_accuracy", "[{:>2}, {:>5.2f}]".format(-1, -1), row["RotAcc"], row["TraAcc"], row["SpcAcc"]), logger)
            for show_idx in show_list:
                print_and_log("{:>16}{:>8}: {:>7.2f}, {:>7.2f}, {:>7.2f}".format(
                    "average_accuracy", "[{:>2}, {:>5.2f}]".format(rot_thresh_list[show_idx], trans_thresh_list[show_idx]),
                    np.sum(rot_acc[:, iter_i, show_idx]) / n * 100, np.sum(trans_acc[:, iter_i, show_idx]) / n * 100,
                    np.sum(space_acc[:, iter_i, show_idx]) / n * 100), logger)
        return {"rot_acc": rot_acc, "trans_acc": trans_acc, "space_acc": space_acc, "overall": overall,
                "num_valid_class": num_valid_class}

    # ------------------------------------------------------------------------------------------------ shared by ADD / ARP-2D
    def _threshold_eval(self, config, all_poses_est, all_poses_gt, error_fn, fixed, curve, curve_scale, area_norm, fmt, header,
                        overall_title, output_dir, pkl_name, logger):
        num_iter = config.TEST.test_iter
        count_all = np.zeros((self.num_classes,), dtype=np.float32)
        count_correct = {k: np.zeros((self.num_classes, num_iter), dtype=np.float32) for k in fixed}
        dx = fmt["dx"]   # the reference passes the python float it built the curve from (:489, :711), not the float32 spacing
        count_correct["mean"] = np.zeros((self.num_classes, num_iter, len(curve)), dtype=np.float32)
        errors = {}
        num_valid_class = 0
        for cls_idx, cls_name in enumerate(self.classes):
            if not self._valid(all_poses_est, all_poses_gt, cls_idx):
                continue
            num_valid_class += 1
            gts = all_poses_gt[cls_idx][0]
            count_all[cls_idx] = len(gts)
            scale = curve_scale(cls_name)
            thr_curve = (curve * np.float32(scale)).astype(np.float32)
            for iter_i in range(num_iter):
                ests = all_poses_est[cls_idx][iter_i]
                err = np.array([error_fn(cls_name, ests[j], gts[j]) for j in range(len(gts))])
                errors[(cls_name, iter_i)] = err
                for k, frac in fixed.items():
                    count_correct[k][cls_idx, iter_i] = np.sum(err < np.float32(frac * scale))
                count_correct["mean"][cls_idx, iter_i] = (err[:, None] < thr_curve[None, :]).sum(0)
        print_and_log(header, logger)
        plot_data = {}
        sums = {k: np.zeros(num_iter) for k in list(fixed) + ["mean"]}
        per_class = {}
        for cls_idx, cls_name in enumerate(self.classes):
            if count_all[cls_idx] == 0:
                continue
            plot_data[cls_name] = []
            for iter_i in range(num_iter):
                print_and_log("** {}, iter {} **".format(cls_name, iter_i + 1), logger)
                y = count_correct["mean"][cls_idx, iter_i] / float(count_all[cls_idx])
                acc_mean = _simps(y, dx=dx) / area_norm * 100
                sums["mean"][iter_i] += acc_mean
                plot_data[cls_name].append((curve.astype(np.float32), y))
                res = {"auc": acc_mean}
                print_and_log("threshold=[0.0, {}], area: {:.2f}".format(fmt["range"], acc_mean), logger)
                for k in fixed:
                    acc = 100 * float(count_correct[k][cls_idx, iter_i]) / float(count_all[cls_idx])
                    sums[k][iter_i] += acc
                    res[k] = acc
                    print_and_log("threshold={}, correct poses: {}, all poses: {}, accuracy: {:.2f}".format(
                        k, count_correct[k][cls_idx, iter_i], count_all[cls_idx], acc), logger)
                per_class[(cls_name, iter_i)] = res
        if output_dir:
            with open(os.path.join(output_dir, pkl_name), "wb") as f:
                pickle.dump(plot_data, f, protocol=2)
        overall = []
        n = max(num_valid_class, 1)
        for iter_i in range(num_iter):
            print_and_log("---------- {} performance over {} classes -----------".format(overall_title, num_valid_class), logger)
            print_and_log("** iter {} **".format(iter_i + 1), logger)
            row = {"auc": sums["mean"][iter_i] / n}
            print_and_log("threshold=[0.0, {}], area: {:.2f}".format(fmt["range"], row["auc"]), logger)
            for k in fixed:
                row[k] = sums[k][iter_i] / n
                print_and_log("threshold={}, mean accuracy: {:.2f}".format(k, row[k]), logger)
            overall.append(row)
        return {"per_class": per_class, "overall": overall, "count_all": count_all, "count_correct": count_correct, "errors": errors,
                "num_valid_class": num_valid_class}

    # ------------------------------------------------------------------------------------------------ :461-681
    def evaluate_pose_add(self, config, all_poses_est, all_poses_gt, output_dir=None, logger=None):
        def err(cls_name, RT, pose_gt):
            fn = adi if cls_name in SYM_CLASSES else add
            return fn(RT[:3, :3], RT[:, 3], pose_gt[:3, :3], pose_gt[:, 3], self._points[cls_name])

        uses_adi = any(c in SYM_CLASSES for c in self.classes)
        return self._threshold_eval(
            config, all_poses_est, all_poses_gt, err, {"0.02": 0.02, "0.05": 0.05, "0.10": 0.10},
            np.arange(0, 0.1, 0.0001).astype(np.float32), lambda c: self._diameters[c], 0.1, {"range": "0.10", "dx": 0.0001},
            "evaluating pose add", "add", output_dir, "{}_xys.pkl".format("adi" if uses_adi else "add"), logger)

    # ------------------------------------------------------------------------------------------------ :683-
    def evaluate_pose_arp_2d(self, config, all_poses_est, all_poses_gt, output_dir=None, logger=None):
        K = np.asarray(config.dataset.INTRINSIC_MATRIX, dtype=np.float64)

        def err(cls_name, RT, pose_gt):
            if cls_name == "eggbox" and re(RT[:3, :3], pose_gt[:3, :3]) > 90:
                RT = se3_mul(RT, RT_Z)
            return arp_2d(RT[:3, :3], RT[:, 3], pose_gt[:3, :3], pose_gt[:, 3], self._points[cls_name], K)

        return self._threshold_eval(
            config, all_poses_est, all_poses_gt, err, {"2": 2.0, "5": 5.0, "10": 10.0, "20": 20.0},
            np.arange(0, 50, 0.1).astype(np.float32), lambda c: 1.0, 50.0, {"range": "50", "dx": 0.1},
            "evaluating pose average re-projection 2d error", "arp_2d", output_dir, "arp_2d_xys.pkl", logger)
