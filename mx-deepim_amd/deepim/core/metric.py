"""Training metrics with the reference's names and normalisations (deepim/core/metric.py:51-140).

The reference reads loss tensors out of the executor's output list; here `update(labels, preds)` takes the dict the HIP module
returns per iteration (`fit_batch` outputs): "flow_loss_sum", "point_matching_loss_sum", "mask_prob", "mask_gt" -- sums are
formed on the device by the loss-gradient kernels, so nothing image-sized is copied to the host for a metric."""
from __future__ import print_function, division

import numpy as np


class EvalMetric(object):
    """mx.metric.EvalMetric surface used by the reference: update / reset / get."""

    def __init__(self, name):
        self.name = name
        self.reset()

    def reset(self):
        self.num_inst = 0
        self.sum_metric = 0.0

    def get(self):
        if self.num_inst == 0:
            return self.name, float("nan")
        return self.name, self.sum_metric / self.num_inst


class CompositeEvalMetric(EvalMetric):
    def __init__(self):
        self.metrics = []
        super(CompositeEvalMetric, self).__init__("composite")

    def add(self, metric):
        self.metrics.append(metric)

    def reset(self):
        for m in getattr(self, "metrics", []):
            m.reset()

    def update(self, labels, preds):
        for m in self.metrics:
            m.update(labels, preds)

    def get(self):
        names, values = [], []
        for m in self.metrics:
            n, v = m.get()
            names.append(n)
            values.append(v)
        return names, values


def _f(x):
    return float(x.item()) if hasattr(x, "item") else float(x)


class Flow_L2LossMetric(EvalMetric):
    def __init__(self, cfg, iter_idx=-1):
        super(Flow_L2LossMetric, self).__init__("Flow_L2Loss")

    def update(self, labels, preds):
        self.sum_metric += _f(preds["flow_loss_sum"])
        self.num_inst += 480 * 640


class Flow_CurLossMetric(EvalMetric):
    def __init__(self, cfg, iter_idx=-1):
        super(Flow_CurLossMetric, self).__init__("Flow_CurLoss")

    def update(self, labels, preds):
        self.sum_metric = _f(preds["flow_loss_sum"])
        self.num_inst = 480 * 640


class Rot_L2LossMetric(EvalMetric):
    """sum of rot_loss = 1 - (rot_gt . rot_est_norm)^2 per update, one instance per update (reference metric.py:80-91)"""

    def __init__(self, cfg, iter_idx=-1):
        super(Rot_L2LossMetric, self).__init__("Rot_L2Loss")

    def update(self, labels, preds):
        self.sum_metric += _f(preds["rot_loss_sum"])
        self.num_inst += 1


class Trans_L2LossMetric(EvalMetric):
    """sum of trans_loss (TRANS_LOSS_TYPE of zoom_trans_est - zoom_trans_gt) per update (reference metric.py:94-105)"""

    def __init__(self, cfg, iter_idx=-1):
        super(Trans_L2LossMetric, self).__init__("Trans_L2Loss")

    def update(self, labels, preds):
        self.sum_metric += _f(preds["trans_loss_sum"])
        self.num_inst += 1


class PointMatchingLossMetric(EvalMetric):
    def __init__(self, cfg, iter_idx=-1):
        super(PointMatchingLossMetric, self).__init__("PointMatchingLoss")
        self.sample_per_iter = cfg["train_iter"]["NUM_3D_SAMPLE"]

    def update(self, labels, preds):
        self.sum_metric += _f(preds["point_matching_loss_sum"])
        self.num_inst += self.sample_per_iter


class MaskLossMetric(EvalMetric):
    def __init__(self, cfg, iter_idx=-1):
        super(MaskLossMetric, self).__init__("MaskLoss")

    def update(self, labels, preds):
        mask_prob, mask_gt = preds["mask_prob"], preds["mask_gt"]
        if hasattr(mask_prob, "cpu"):
            mask_prob, mask_gt = mask_prob.cpu().numpy(), mask_gt.cpu().numpy()
        mask_loss = -(mask_gt * np.log(mask_prob + 1e-19) + (1 - mask_gt) * np.log(1 - mask_prob + 1e-19))
        self.sum_metric += np.sum(mask_loss)
        self.num_inst += 480 * 640
