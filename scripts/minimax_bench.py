#!/usr/bin/env python3
"""Time tron_minimax_actions (depth-2 Voronoi search for one player of every env).
Usage: python scripts/minimax_bench.py [--envs N --width W --steps K --mode voronoi|distwall]"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "deep-q-learning_tron_amd"))

import torch  # noqa: E402
from tron.vec import VecTron  # noqa: E402


def run(n, w, pre_steps, mode, reps):
    env = VecTron(n, w, seed=1, obs_format="codes")
    env.reset()
    for _ in range(pre_steps):                    # both players follow the search: boards fill up like real games
        a = torch.stack([env.minimax_actions(1), env.minimax_actions(2)], 1)
        env.step(a, autoreset=True)
    out = torch.empty(n, dtype=torch.int8, device="cuda")
    for _ in range(3):
        env.minimax_actions(2, mode, out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        env.minimax_actions(2, mode, out=out)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    filled = float((env.obs[:, 0, 1:-1, 1:-1] != 1).float().mean())
    print(f"N={n:6d} W={w:2d} mode={mode:8s} after {pre_steps:3d} searched steps (board {100 * filled:4.1f}% full): "
          f"{ms * 1e3:9.1f} us/launch  {n / ms / 1e3:8.2f} M moves/s", flush=True)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", type=int)
    ap.add_argument("--width", type=int)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--mode", default="voronoi")
    ap.add_argument("--reps", type=int, default=10)
    a = ap.parse_args()
    if a.envs:
        run(a.envs, a.width or 10, a.steps or 0, a.mode, a.reps)
    else:
        for n, w in ((4096, 10), (65536, 24), (16384, 32)):
            for k in (0, w // 2, 2 * w):
                run(n, w, k, "voronoi", a.reps)
            run(n, w, w // 2, "distwall", a.reps)
