"""Learning-rate schedule of the reference training driver (lib/utils/lr_scheduler.py:45-70, deepim/train.py:318-332).

`WarmupMultiFactorScheduler` keeps MXNet's LRScheduler calling convention: the optimizer sets `base_lr` and calls the
object with `num_update` (the number of optimizer steps so far) to get that step's learning rate."""
from __future__ import print_function, division

import logging


class LRScheduler(object):
    """mxnet.lr_scheduler.LRScheduler protocol: the optimizer writes its learning rate into `base_lr`, then asks
    `scheduler(num_update)` for the rate of update number `num_update`."""

    def __init__(self, base_lr=0.01):
        self.base_lr = base_lr

    def __call__(self, num_update):
        raise NotImplementedError("must override this")


def _check_milestones(step, factor):
    assert isinstance(step, list) and len(step) >= 1
    if any(s < 1 for s in step):
        raise ValueError("Schedule step must be greater or equal than 1 round")
    if any(b <= a for a, b in zip(step, step[1:])):
        raise ValueError("Schedule step must be an increasing integer list")
    if factor > 1.0:
        raise ValueError("Factor must be no more than 1 to make lr reduce")


class WarmupMultiFactorScheduler(LRScheduler):
    """Piecewise-constant decay: every milestone in `step` that the update counter has gone PAST (num_update > milestone)
    multiplies `base_lr` by `factor`, once.  While `warmup` and num_update < warmup_step the answer is `warmup_lr` and no
    milestone is consumed.  State the callers of the reference read is kept under the same names: `step`, `factor`,
    `cur_step_ind` (milestones consumed so far), `count` (the last consumed milestone), `base_lr` (current rate)."""

    def __init__(self, step, factor=1, warmup=False, warmup_lr=0, warmup_step=0):
        super(WarmupMultiFactorScheduler, self).__init__()
        _check_milestones(step, factor)
        self.step, self.factor = step, factor
        self.warmup, self.warmup_lr, self.warmup_step = warmup, warmup_lr, warmup_step
        self.cur_step_ind = 0
        self.count = 0

    def _pending(self, num_update):
        """milestones not yet consumed that num_update has passed -- usually 0 or 1; several when a run resumes far ahead"""
        n = 0
        while self.cur_step_ind + n < len(self.step) and num_update > self.step[self.cur_step_ind + n]:
            n += 1
        return n

    def __call__(self, num_update):
        if self.warmup and num_update < self.warmup_step:
            return self.warmup_lr
        for _ in range(self._pending(num_update)):
            self.count = self.step[self.cur_step_ind]
            self.cur_step_ind += 1
            self.base_lr *= self.factor
            logging.info("Update[%d]: Change learning rate to %0.5e", num_update, self.base_lr)
        return self.base_lr


def build_lr_schedule(base_lr, lr_step, begin_epoch, num_pairs, batch_size, warmup=False, warmup_lr=0, warmup_step=0, lr_factor=0.1):
    """deepim/train.py:318-332: epochs already done (resume at begin_epoch) are folded into the starting lr, the rest become
    iteration thresholds.  -> (lr, scheduler) with scheduler.base_lr = lr like mx.optimizer does at creation."""
    lr_epoch = [float(epoch) for epoch in str(lr_step).split(",")]
    lr_epoch_diff = [epoch - begin_epoch for epoch in lr_epoch if epoch > begin_epoch]
    lr = base_lr * (lr_factor ** (len(lr_epoch) - len(lr_epoch_diff)))
    lr_iters = [int(epoch * num_pairs / batch_size) for epoch in lr_epoch_diff]
    if not lr_iters:  # every step already passed: constant lr (MXNet's fit would get an empty list and the assert above)
        lr_iters = [1 << 62]
    sched = WarmupMultiFactorScheduler(lr_iters, lr_factor, warmup, warmup_lr, warmup_step)
    sched.base_lr = lr
    return lr, sched
