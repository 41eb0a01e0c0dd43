"""Named wall-clock timers - counterpart of lib/utils/timer.py (``Timer().tic(name)`` / ``toc(name, average)``,
``average_time`` / ``total_time``; ``lib/model/test.py:170-203`` and ``lib/model/train_val.py:356-361`` keep them in
dicts and assign them to ``net.timers``).  Like the reference, a reading is taken after the device has drained, so a
``tic``/``toc`` pair around asynchronous kernel launches measures the kernels and not the launch calls."""
import time
from collections import defaultdict

import torch


def _now():
    if torch.cuda.is_available():
        torch.cuda.synchronize()
    return time.time()


class Timer(object):
    def __init__(self):
        self._t0 = {}
        self._last = {}
        self._sum = defaultdict(float)
        self._n = defaultdict(int)

    def tic(self, name='default'):
        self._t0[name] = _now()

    def toc(self, name='default', average=True):
        dt = _now() - self._t0[name]
        self._last[name] = dt
        self._sum[name] += dt
        self._n[name] += 1
        return self._sum[name] / self._n[name] if average else dt

    def average_time(self, name='default'):
        return self._sum[name] / self._n[name]

    def total_time(self, name='default'):
        return self._sum[name]


timer = Timer()
