#!/usr/bin/env python3
"""Kernel-variant sweep on one GPU: (waves per tile, envs per tile) x {full step, encode only,
step without obs}.  Development tool; bench.py is the judged entry point."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "deep-q-learning_tron_amd")]

import torch  # noqa: E402
from tron.vec import VecTron  # noqa: E402


def timeit(fn, iters):
    for _ in range(10):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3   # us


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", type=int, default=65536)
    ap.add_argument("--width", type=int, default=24)
    ap.add_argument("--iters", type=int, default=200)
    ap.add_argument("--obs", default="codes")
    ap.add_argument("--nw", default="4")
    ap.add_argument("--tile", default="16,32,64", help="envs per workgroup tile (TRON_TILE_ENVS)")
    ap.add_argument("--extra-env", default="", help="KEY=VAL,... extra env vars for the native lib")
    a = ap.parse_args()
    for kv in filter(None, a.extra_env.split(",")):
        k, v = kv.split("=")
        os.environ[k] = v
    G = (a.width + 2) ** 2
    balg = 3 * G + 32
    print(f"N={a.envs} W={a.width} obs={a.obs}  alg bytes/env-step={balg}")
    print(f"{'NW':>3} {'E':>3} {'full_us':>9} {'TB/s':>6} {'frac':>6} {'enc_us':>8} {'step_us':>8}")
    for nw in [int(x) for x in a.nw.split(",")]:
        for e in [int(x) for x in a.tile.split(",")]:
            os.environ["TRON_TILE_ENVS"] = str(e)
            env = VecTron(a.envs, a.width, seed=0x5EED, obs_format=a.obs)
            env.reset()
            full = timeit(env.step_fn(autoreset=True), a.iters)
            enc = timeit(lambda: env.encode(), a.iters)
            env2 = VecTron(a.envs, a.width, seed=0x5EED, obs_format=None)
            env2.reset()
            st = timeit(env2.step_fn(autoreset=True), a.iters)
            tbs = balg * a.envs / (full * 1e-6) / 1e12
            print(f"{nw:>3} {e:>3} {full:9.2f} {tbs:6.2f} {tbs/8:6.3f} {enc:8.2f} {st:8.2f}", flush=True)
            env.close()
            env2.close()


if __name__ == "__main__":
    main()
