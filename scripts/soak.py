#!/usr/bin/env python3
"""Long-run parity soak: K rollout steps of N envs on the GPU, then the first P envs' full state against
the oracle stepped K times on the same Philox key (envs are keyed by index, so a prefix is comparable).
Usage: python scripts/soak.py [--envs 65536 --width 24 --steps 20000 --prefix 2048 --mode none]"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "deep-q-learning_tron_amd")]
import numpy as np  # noqa: E402
import torch  # noqa: E402
import oracle  # noqa: E402
from tron.vec import VecTron  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--envs", type=int, default=65536)
ap.add_argument("--width", type=int, default=24)
ap.add_argument("--steps", type=int, default=20000)
ap.add_argument("--prefix", type=int, default=2048)
ap.add_argument("--mode", default="none")
ap.add_argument("--nonreversing", action="store_true")
a = ap.parse_args()
mode = None if a.mode == "none" else a.mode
env = VecTron(a.envs, a.width, mode=mode, seed=0xABCDE, rank=4, obs_format="codes")
ref = oracle.VecOracle(a.prefix, a.width, mode=mode, seed=0xABCDE, stream=4)
env.reset()
ref.reset_all()
totals = torch.zeros(4, dtype=torch.int64, device="cuda")
side = torch.cuda.Stream()
torch.cuda.synchronize()              # reset ran on the default stream; side streams do not wait for it
t0 = time.perf_counter()
with torch.cuda.stream(side):
    env.rollout_random(a.steps, totals, nonreversing=a.nonreversing)
torch.cuda.synchronize()
t_gpu = time.perf_counter() - t0
oracle.set_threads(min(16, len(os.sched_getaffinity(0))))
t0 = time.perf_counter()
for k in range(a.steps):
    o, _, _, _ = ref.step(autoreset=True, nonreversing=a.nonreversing, want_obs=(k == a.steps - 1))
t_cpu = time.perf_counter() - t0
P = a.prefix
st = {k: v.cpu().numpy() for k, v in env.state().items()}
ok = (np.array_equal(env.grid().cpu().numpy().reshape(a.envs, -1)[:P], ref.grid) and np.array_equal(st["pos"][:P], ref.pos)
      and np.array_equal(st["alive"][:P], ref.alive) and np.array_equal(st["dir"][:P], ref.dir)
      and np.array_equal(st["weight"][:P], ref.weight) and np.array_equal(st["degree"][:P], ref.degree)
      and np.array_equal(st["counters"][:P, 0].astype(np.uint32), ref.tick)
      and np.array_equal(st["counters"][:P, 1].astype(np.uint32), ref.episode)
      and np.array_equal(st["counters"][:P, 2].astype(np.uint32), ref.eplen)
      and np.array_equal(env.obs.cpu().numpy().reshape(a.envs, 2, -1)[:P], o))
tot = totals.cpu().numpy()
print(f"{a.envs} x {a.width}x{a.width} mode={a.mode} {a.steps} steps: GPU {t_gpu:.2f} s ({a.envs * a.steps / t_gpu / 1e9:.2f} G env-steps/s), "
      f"oracle prefix {P} in {t_cpu:.1f} s; totals steps={tot[0]} p1={tot[1]} p2={tot[2]} draws={tot[3]}; "
      f"games/env {ref.episode.mean():.0f}; prefix state+obs equal: {ok}", flush=True)
assert tot[0] == a.envs * a.steps and ok
