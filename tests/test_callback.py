"""deepim/core/callback.py Speedometer against lines the REFERENCE's Speedometer printed (tests/golden/callback_golden.json, generated
by tests/golden/make_golden.py importing /root/reference/deepim/core/callback.py under a scripted clock): same events, same clock
ticks, the same line (or no line) per callback -- the protocol of the reference's train loop (deepim/core/module.py:1036-1049)."""
import json
import os

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


class _Metric(object):
    def __init__(self, names, values):
        self.names, self.values = names, values

    def get(self):
        return self.names, self.values


@pytest.mark.parametrize("case", range(4))
def test_speedometer_matches_reference_lines(case):
    from deepim.core.callback import BatchEndParam, Speedometer

    script = json.load(open(os.path.join(HERE, "golden", "callback_golden.json")))[case]
    ticks = iter(script["clock"])
    sp = Speedometer(script["batch_size"], script["frequent"], clock=lambda: next(ticks))
    n_lines = 0
    for ev, want in zip(script["events"], script["lines"]):
        metric = _Metric(script["metric_names"], ev["metric"]) if ev["metric"] is not None else None
        got = sp(BatchEndParam(ev["epoch"], ev["nbatch"], metric, None))
        assert got == want, (ev, got, want)
        n_lines += want is not None
    assert n_lines >= 3   # the script exercises the logging branch
    # the clock is consumed exactly as the reference consumes time.time(): one tick per silent (re)start, two per logged line
    used = script["clock"].index(next(ticks))
    assert used == sum(1 for w in script["lines"] if w is not None) * 2 + _restarts(script)


def _restarts(script):
    """callbacks that only (re)open the window: the first one, and each one whose batch counter went backwards"""
    n, prev = 0, None
    for ev in script["events"]:
        if prev is None or ev["nbatch"] < prev:
            n += 1
        prev = ev["nbatch"]
    return n
