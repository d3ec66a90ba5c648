"""Scalar logging for the trainers.  The reference logs through
torch.utils.tensorboard.SummaryWriter (DDQN.py:207,342-344; ACKTR.py:185-188,401-421);
tensorboard is not part of this image, so scalars always go to a JSON-lines file and are
mirrored to tensorboard only when it can be imported."""
import json
import os
import time


class ScalarWriter:
    def __init__(self, logdir):
        os.makedirs(logdir, exist_ok=True)
        self.path = os.path.join(logdir, "scalars.jsonl")
        self._f = open(self.path, "a")
        try:
            from torch.utils.tensorboard import SummaryWriter
            self._tb = SummaryWriter(logdir)
        except Exception:
            self._tb = None

    def add_scalar(self, tag, value, step):
        rec = {"t": time.time(), "tag": tag, "value": float(value), "step": int(step)}
        self._f.write(json.dumps(rec) + "\n")
        self._f.flush()
        if self._tb is not None:
            self._tb.add_scalar(tag, float(value), int(step))

    def close(self):
        self._f.close()
        if self._tb is not None:
            self._tb.close()


def read_scalars(path):
    with open(path) as f:
        return [json.loads(line) for line in f if line.strip()]
