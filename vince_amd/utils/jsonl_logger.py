"""Scalar logger with the two methods of dg_util's tensorboard_logger.Logger that the hot path calls
(reference solvers/vince_solver.py:511-512 `dict_log`; solvers/base_solver.py:121-128 `scalar_summary`): one JSON line per
call in <log_dir>/events.jsonl.  solver_runner uses it when dg_util (an unpinned dependency of the reference,
requirements.txt:21) is not installed; image / histogram summaries are accepted and dropped."""
import json
import os


class Logger:
    def __init__(self, log_dir):
        self.log_dir = log_dir
        os.makedirs(log_dir, exist_ok=True)
        self._f = open(os.path.join(log_dir, "events.jsonl"), "a")

    def _write(self, rec):
        self._f.write(json.dumps(rec) + "\n")
        self._f.flush()

    def dict_log(self, scalars, step):
        self._write({"step": int(step), "scalars": {k: float(v) for k, v in scalars.items()}})

    def scalar_summary(self, tag, value, step=None, increment_counter=False):
        self._write({"step": None if step is None else int(step), "scalars": {tag: float(value)}})

    def image_summary(self, *a, **k):
        pass

    def network_conv_summary(self, *a, **k):
        pass

    def histo_summary(self, *a, **k):
        pass
