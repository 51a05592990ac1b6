"""WarmupMultiFactorScheduler / build_lr_schedule (reference lib/utils/lr_scheduler.py:45-70, deepim/train.py:318-332),
Speedometer and metric normalisations."""
import numpy as np
import pytest

from lib.utils.lr_scheduler import WarmupMultiFactorScheduler, build_lr_schedule


def test_multifactor_steps_and_warmup():
    s = WarmupMultiFactorScheduler([10, 20], factor=0.1, warmup=True, warmup_lr=1e-5, warmup_step=4)
    s.base_lr = 1e-4
    got = [s(n) for n in (0, 3, 4, 10, 11, 20, 21, 1000)]
    np.testing.assert_allclose(got, [1e-5, 1e-5, 1e-4, 1e-4, 1e-5, 1e-5, 1e-6, 1e-6], rtol=1e-12)
    assert s.cur_step_ind == 2 and s.count == 20


def test_resume_jumps_over_several_steps_at_once():
    s = WarmupMultiFactorScheduler([5, 8, 12], factor=0.5)
    s.base_lr = 1.0
    assert s(100) == pytest.approx(0.125)  # the `while` in __call__ (reference :57)


def test_argument_checks():
    with pytest.raises(ValueError):
        WarmupMultiFactorScheduler([5, 5], 0.1)
    with pytest.raises(ValueError):
        WarmupMultiFactorScheduler([0], 0.1)
    with pytest.raises(ValueError):
        WarmupMultiFactorScheduler([5], 1.5)
    with pytest.raises(AssertionError):
        WarmupMultiFactorScheduler([], 0.1)


def test_build_schedule_like_train_py():
    # shipped yaml: lr 1e-4, lr_step '4,6', 8 epochs; 4000 pairs, batch 16
    lr, s = build_lr_schedule(1e-4, "4, 6", begin_epoch=0, num_pairs=4000, batch_size=16)
    assert lr == pytest.approx(1e-4) and s.step == [1000, 1500] and s.base_lr == pytest.approx(1e-4)
    # resume at epoch 5: first step already applied, one threshold left, counted from the resume point
    lr, s = build_lr_schedule(1e-4, "4, 6", begin_epoch=5, num_pairs=4000, batch_size=16)
    assert lr == pytest.approx(1e-5) and s.step == [250]
    assert s(251) == pytest.approx(1e-6)
    lr, s = build_lr_schedule(1e-4, "4, 6", begin_epoch=7, num_pairs=4000, batch_size=16)
    assert lr == pytest.approx(1e-6) and s(10 ** 6) == pytest.approx(1e-6)


def test_speedometer_and_metrics():
    from deepim.core.callback import BatchEndParam, Speedometer
    from deepim.core import metric
    from scene import make_train_config

    cfg = make_train_config()
    comp = metric.CompositeEvalMetric()
    for m in (metric.Flow_L2LossMetric(cfg), metric.Flow_CurLossMetric(cfg), metric.PointMatchingLossMetric(cfg), metric.MaskLossMetric(cfg)):
        comp.add(m)
    prob = np.full((1, 1, 480, 640), 0.25, np.float32)
    gt = np.zeros((1, 1, 480, 640), np.float32)
    for _ in range(2):
        comp.update(None, {"flow_loss_sum": 307200.0, "point_matching_loss_sum": 3.0 * cfg.train_iter.NUM_3D_SAMPLE, "mask_prob": prob, "mask_gt": gt})
    names, values = comp.get()
    assert names == ["Flow_L2Loss", "Flow_CurLoss", "PointMatchingLoss", "MaskLoss"]
    np.testing.assert_allclose(values, [1.0, 1.0, 3.0, -np.log(0.75)], rtol=1e-5)
    sp = Speedometer(batch_size=16, frequent=2)
    assert sp(BatchEndParam(0, 1, comp, None)) is None          # arms the timer
    line = sp(BatchEndParam(0, 2, comp, None))
    assert line.startswith("Epoch[0] Batch [2]\tSpeed: ") and "Train-Flow_L2Loss=1.000000" in line and "MaskLoss=" in line
