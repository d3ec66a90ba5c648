"""Scalar logging for the trainers.  The reference logs through
torch.utils.tensorboard.SummaryWriter (DDQN.py:207,342-344; ACKTR.py:185-188,401-421);
tensorboard is not part of this image, so every scalar goes to (1) a TensorBoard event file written
by tron/tbevents.py (same on-disk format, `tensorboard --logdir` reads it) and (2) a JSON-lines file
that needs no tooling at all."""
import json
import os
import time

from .tbevents import EventFileWriter


class ScalarWriter:
    def __init__(self, logdir):
        os.makedirs(logdir, exist_ok=True)
        self.path = os.path.join(logdir, "scalars.jsonl")
        self._f = open(self.path, "a")
        self._tb = EventFileWriter(logdir)
        self.events_path = self._tb.path

    def add_scalar(self, tag, value, step):
        now = time.time()
        rec = {"t": now, "tag": tag, "value": float(value), "step": int(step)}
        self._f.write(json.dumps(rec) + "\n")
        self._f.flush()
        self._tb.add_scalar(tag, float(value), int(step), now)

    def close(self):
        self._f.close()
        self._tb.close()


def read_scalars(path):
    with open(path) as f:
        return [json.loads(line) for line in f if line.strip()]
