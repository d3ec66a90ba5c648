#!/usr/bin/env python3
"""Where one ACKTR iteration at BASELINE config 5 spends its time: wall seconds (synchronised) inside KFACOptimizer.step (with the
eigendecompositions separately), inside the statistics hooks (input / gradient factors) and in the rest of Brain.update.
usage: acktr_update_breakdown.py [envs] [width] [iterations]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "deep-q-learning_tron_amd")]
import config, torch  # noqa
import ACKTR
from Net import kfac

envs = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
width = int(sys.argv[2]) if len(sys.argv) > 2 else 32
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 1
T = {}


def timed(name, fn):
    def wrapper(*a, **k):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        r = fn(*a, **k)
        torch.cuda.synchronize()
        T[name] = T.get(name, 0.0) + time.perf_counter() - t0
        return r
    return wrapper


ACKTR.train(n_envs=envs, width=width, model="mul", reward="3", iterations=1, acktr=True, log_every=0)     # warm-up
kfac.KFACOptimizer.step = timed("optimizer.step", kfac.KFACOptimizer.step)
kfac.cov_inputs = timed("cov_inputs", kfac.cov_inputs)
kfac.cov_grads = timed("cov_grads", kfac.cov_grads)
_eigh = torch.linalg.eigh
torch.linalg.eigh = timed("eigh (inside step)", _eigh)
ACKTR.Brain.update = timed("Brain.update", ACKTR.Brain.update)
torch.cuda.synchronize()
t0 = time.perf_counter()
o = ACKTR.train(n_envs=envs, width=width, model="mul", reward="3", iterations=iters, acktr=True, log_every=0)
torch.cuda.synchronize()
wall = time.perf_counter() - t0
print(f"{iters} iteration(s), {envs} envs x {width}x{width}: {wall:.2f} s wall (timers synchronise: slower than bench.py --acktr)")
for k, v in sorted(T.items(), key=lambda kv: -kv[1]):
    print(f"  {k:24s} {v / iters:7.3f} s per iteration")
