"""The evaluation tables of /root/reference/lib/dataset/LM6D_REFINE.py (evaluate_pose :329-459, evaluate_pose_add :461-681,
evaluate_pose_arp_2d :683-893), restated loop for loop -- per pose, per threshold -- and returning the arrays the reference only
prints / plots.  Oracle; test infrastructure only: tests/test_evaluation.py compares lib/dataset/evaluation.PoseEvaluator (which
computes the same tables vectorised) against this.  Primitives: oracle/pose_error.py (pinned by tests/golden/pose_error_golden.npz)
and oracle/se3.py (pinned by tests/golden/se3_golden.npz).  LM6D_REFINE.py itself needs cv2 / six / tqdm to import, none of which
this image has, so the tables have no reference-generated vectors: "parity unpinned" for the aggregation, pinned for its inputs."""
import numpy as np

from . import pose_error as pe
from . import se3

RT_Z = np.array([[-1, 0, 0, 0], [0, -1, 0, 0], [0, 0, 1, 0]])  # the eggbox twin (:357, :738)


def _valid(est, gt, ci):
    return bool(est[ci][0] and gt[ci][0])  # :342, :502, :727


def evaluate_pose(classes, est, gt, num_iter):
    """:329-459 -> rot_acc, trans_acc, space_acc [class, iter, 10], overall rows (:432-448), num_valid_class"""
    rot_t, tr_t = np.arange(1, 11, 1), np.arange(0.01, 0.11, 0.01)
    nm = len(rot_t)
    rot_acc, trans_acc, space_acc = (np.zeros((len(classes), num_iter, nm)) for _ in range(3))
    nvalid = 0
    for ci, name in enumerate(classes):
        if not _valid(est, gt, ci):
            continue
        nvalid += 1
        for it in range(num_iter):
            gts, ests = gt[ci][0], est[ci][it]
            rr, tt = np.zeros((len(gts), 1)), np.zeros((len(gts), 1))
            for j in range(len(gts)):
                r, t = se3.calc_rt_dist_m(ests[j], gts[j])
                if name == "eggbox" and r > 90:
                    r, t = se3.calc_rt_dist_m(se3.se3_mul(ests[j], RT_Z), gts[j])
                rr[j, 0], tt[j, 0] = r, t
            for k in range(nm):
                rot_acc[ci, it, k] = np.mean(rr < rot_t[k])
                trans_acc[ci, it, k] = np.mean(tt < tr_t[k])
                space_acc[ci, it, k] = np.mean(np.logical_and(rr < rot_t[k], tt < tr_t[k]))
    overall = [{"RotAcc": np.sum(rot_acc[:, it, :]) / (nvalid * nm) * 100, "TraAcc": np.sum(trans_acc[:, it, :]) / (nvalid * nm) * 100,
                "SpcAcc": np.sum(space_acc[:, it, :]) / (nvalid * nm) * 100} for it in range(num_iter)]
    return rot_acc, trans_acc, space_acc, overall, nvalid


def _simps(y, dx):
    """scipy.integrate.simps (removed in scipy 1.14; `simpson` is the same rule) as :565-571 / :794-797 call it"""
    from scipy import integrate

    return (integrate.simpson if hasattr(integrate, "simpson") else integrate.simps)(y, dx=dx)


def _count_loop(classes, est, gt, num_iter, error_of, fixed_thresholds, curve):
    """the counting loops of :500-541 / :725-772: per pose, per fixed threshold, per curve threshold"""
    n = len(classes)
    count_all = np.zeros((n,), np.float32)
    cc = {k: np.zeros((n, num_iter), np.float32) for k in fixed_thresholds}
    cc["mean"] = np.zeros((n, num_iter, curve.shape[-1]), np.float32)
    nvalid = 0
    for ci, name in enumerate(classes):
        if not _valid(est, gt, ci):
            continue
        nvalid += 1
        for it in range(num_iter):
            gts, ests = gt[ci][0], est[ci][it]
            for j in range(len(gts)):
                if it == 0:
                    count_all[ci] += 1
                err = error_of(name, ests[j], gts[j])
                for k, thr in fixed_thresholds.items():
                    if err < thr[ci, it]:
                        cc[k][ci, it] += 1
                for ti in range(curve.shape[-1]):
                    if err < curve[ci, it, ti]:
                        cc["mean"][ci, it, ti] += 1
    return count_all, cc, nvalid


def _accuracies(classes, num_iter, count_all, cc, keys, dx, area_norm, nvalid):
    """:553-590 / :781-822 per class, :630-677 / :870-893 over classes"""
    per_class, sums = {}, {k: np.zeros(num_iter) for k in list(keys) + ["auc"]}
    for ci, name in enumerate(classes):
        if count_all[ci] == 0:
            continue
        for it in range(num_iter):
            res = {"auc": _simps(cc["mean"][ci, it] / float(count_all[ci]), dx) / area_norm * 100}
            for k in keys:
                res[k] = 100 * float(cc[k][ci, it]) / float(count_all[ci])
            for k in res:
                sums[k][it] += res[k]
            per_class[(name, it)] = res
    overall = [{k: sums[k][it] / nvalid for k in sums} for it in range(num_iter)]
    return per_class, overall


def evaluate_pose_add(classes, points, diameters, est, gt, num_iter):
    """:461-681 -> per_class {(class, iter): {"auc", "0.02", "0.05", "0.10"}}, overall rows, count_correct, count_all"""
    n, dx = len(classes), 0.0001
    thr = {k: np.zeros((n, num_iter), np.float32) for k in ("0.02", "0.05", "0.10")}
    curve = np.tile(np.arange(0, 0.1, dx).astype(np.float32), (n, num_iter, 1))
    for i, name in enumerate(classes):  # :494-498
        thr["0.02"][i, :] = 0.02 * diameters[name]
        thr["0.05"][i, :] = 0.05 * diameters[name]
        thr["0.10"][i, :] = 0.10 * diameters[name]
        curve[i, :, :] *= diameters[name]

    def error_of(name, RT, g):  # :516-534
        fn = pe.adi if name in ("eggbox", "glue", "bowl", "cup") else pe.add
        return fn(RT[:3, :3], RT[:, 3], g[:3, :3], g[:, 3], points[name])

    count_all, cc, nvalid = _count_loop(classes, est, gt, num_iter, error_of, thr, curve)
    per_class, overall = _accuracies(classes, num_iter, count_all, cc, ("0.02", "0.05", "0.10"), dx, 0.1, nvalid)
    return per_class, overall, cc, count_all


def evaluate_pose_arp_2d(classes, points, K, est, gt, num_iter):
    """:683-893 -> per_class {(class, iter): {"auc", "2", "5", "10", "20"}}, overall rows, count_correct, count_all"""
    n, dx = len(classes), 0.1
    thr = {k: np.full((n, num_iter), float(k), np.float32) for k in ("2", "5", "10", "20")}  # :719-723
    curve = np.tile(np.arange(0, 50, dx).astype(np.float32), (n, num_iter, 1))

    def error_of(name, RT, g):  # :736-760
        if name == "eggbox" and pe.re(RT[:3, :3], g[:3, :3]) > 90:
            RT = se3.se3_mul(RT, RT_Z)
        return pe.arp_2d(RT[:3, :3], RT[:, 3], g[:3, :3], g[:, 3], points[name], K)

    count_all, cc, nvalid = _count_loop(classes, est, gt, num_iter, error_of, thr, curve)
    per_class, overall = _accuracies(classes, num_iter, count_all, cc, ("2", "5", "10", "20"), dx, 50.0, nvalid)
    return per_class, overall, cc, count_all


# ---------------------------------------------------------------------------------------------------------------------
# Test-time flow error (deepim/core/tester.py:500-512, :675-736).  PINNED: tests/golden/epe_golden.npz holds the outputs of the
# reference's own calc_EPE_one_pair (executed from its file by tests/golden/make_golden.py) on the reference's own calc_flow.
def flow_gt_list(flow, visible, depth_rendered):
    """the list par_generate_gt returns under "flow" (tester.py:706-716): [flow_i2r, visible, visible == 0 and depth_rendered == 0]"""
    return [flow, visible, np.logical_and(visible == 0, depth_rendered == 0)]


def calc_EPE_one_pair(flow_pred_list, flow_gt, flow_type="flow"):
    """tester.py:719-736; flow_pred_list[flow_type] is the (H,W,2) float16 array of tester.py:485-487"""
    pred = flow_pred_list[flow_type]
    gt, visible, bg = flow_gt[flow_type][0], flow_gt[flow_type][1], flow_gt[flow_type][2]
    dx = gt[:, :, 0] - pred[:, :, 0]
    dy = gt[:, :, 1] - pred[:, :, 1]
    dist = np.sqrt(np.square(dx) + np.square(dy))
    either = np.logical_or(visible, bg)
    return {"epe_all": dist.sum(), "num_all": dist.size, "epe_viz": dist[visible == 1].sum(), "num_viz": visible.sum(),
            "epe_vizbg": dist[either].sum(), "num_vizbg": either.sum()}


def epe_of_batch(flow_est_crop, flow, visible, depth_rendered):
    """(B,2,H,W) network output + (B,2,H,W) / (B,1,H,W) / (B,1,H,W) labels -> (B,5) [epe_all, epe_viz, epe_vizbg, num_viz, num_vizbg],
    through the float16 store of tester.py:485-487 (`output.asnumpy().transpose((2, 3, 1, 0))` squeezed: (H,W,2) per sample)"""
    out = []
    for b in range(flow_est_crop.shape[0]):
        cur = {"flow": np.asarray(flow_est_crop[b]).transpose(1, 2, 0).astype("float16")}
        gt = flow_gt_list(np.asarray(flow[b]).transpose(1, 2, 0), np.asarray(visible[b, 0]), np.asarray(depth_rendered[b, 0]))
        r = calc_EPE_one_pair(cur, {"flow": gt})
        out.append([r["epe_all"], r["epe_viz"], r["epe_vizbg"], r["num_viz"], r["num_vizbg"]])
    return np.array(out, dtype=np.float64)
