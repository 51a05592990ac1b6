"""Batch-end callback of the reference (deepim/core/callback.py:10-52): samples/sec + the running metrics every `frequent` batches."""
from __future__ import print_function, division

import collections
import logging
import time

BatchEndParam = collections.namedtuple("BatchEndParams", ["epoch", "nbatch", "eval_metric", "locals"])


class Speedometer(object):
    def __init__(self, batch_size, frequent=50):
        self.batch_size = batch_size
        self.frequent = frequent
        self.init = False
        self.tic = 0
        self.last_count = 0

    def __call__(self, param):
        count = param.nbatch
        if self.last_count > count:
            self.init = False
        self.last_count = count
        if self.init:
            if count % self.frequent == 0:
                speed = self.frequent * self.batch_size / (time.time() - self.tic)
                if param.eval_metric is not None:
                    name, value = param.eval_metric.get()
                    s = "Epoch[%d] Batch [%d]\tSpeed: %.2f samples/sec\tTrain-" % (param.epoch, count, speed)
                    for n, v in zip(name, value):
                        s += "%s=%f,\t" % (n, v)
                else:
                    s = "Iter[%d] Batch [%d]\tSpeed: %.2f samples/sec" % (param.epoch, count, speed)
                logging.info(s)
                print(s)
                self.tic = time.time()
                return s
        else:
            self.init = True
            self.tic = time.time()
        return None
