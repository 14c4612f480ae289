"""RollingAverageMeter / AverageMeter (dg_util.python_utils.average_meter stand-ins):
``.update(val[, n])``, ``.val``, ``.avg``."""
from collections import deque


class RollingAverageMeter:
    def __init__(self, window):
        self.window = max(int(window), 1)
        self.values = deque(maxlen=self.window)
        self.val = 0.0

    def update(self, val, n=1):
        val = float(val)
        self.val = val
        self.values.append(val)

    @property
    def avg(self):
        return sum(self.values) / max(len(self.values), 1)


class AverageMeter:
    def __init__(self):
        self.sum, self.count, self.val = 0.0, 0, 0.0

    def update(self, val, n=1):
        val = float(val)
        self.val = val
        self.sum += val * n
        self.count += n

    @property
    def avg(self):
        return self.sum / max(self.count, 1)
