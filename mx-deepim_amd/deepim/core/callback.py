"""Batch-end throughput logger with the reference's name and call protocol (deepim/core/callback.py:10-52).

`Speedometer(batch_size, frequent)(BatchEndParam(epoch, nbatch, eval_metric, locals))` logs one line every `frequent`
batches -- "Epoch[e] Batch [n]\tSpeed: x samples/sec\tTrain-<metric>=<value>,\t..." -- and restarts its clock whenever the
batch counter goes backwards (a new epoch).  Written as a small window timer; the returned string is what was logged."""
from __future__ import print_function, division

import logging
import time
from collections import namedtuple

BatchEndParam = namedtuple("BatchEndParams", ["epoch", "nbatch", "eval_metric", "locals"])


class Speedometer(object):
    def __init__(self, batch_size, frequent=50, clock=time.time):
        self.batch_size = int(batch_size)
        self.frequent = int(frequent)
        self._clock = clock
        self._window_start = None   # None: no window open (first call, or the batch counter was reset)
        self._prev_nbatch = -1

    def _format(self, param, speed):
        metric = param.eval_metric
        if metric is None:
            return "Iter[{:d}] Batch [{:d}]\tSpeed: {:.2f} samples/sec".format(param.epoch, param.nbatch, speed)
        names, values = metric.get()
        head = "Epoch[{:d}] Batch [{:d}]\tSpeed: {:.2f} samples/sec\tTrain-".format(param.epoch, param.nbatch, speed)
        return head + "".join("{}={:f},\t".format(n, v) for n, v in zip(names, values))

    def __call__(self, param):
        nbatch = param.nbatch
        went_back = nbatch < self._prev_nbatch
        self._prev_nbatch = nbatch
        if self._window_start is None or went_back:
            self._window_start = self._clock()
            return None
        if nbatch % self.frequent:
            return None
        now = self._clock()
        line = self._format(param, self.frequent * self.batch_size / max(now - self._window_start, 1e-12))
        self._window_start = self._clock()
        logging.info(line)
        print(line)
        return line
