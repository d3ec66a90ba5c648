#!/usr/bin/env python3
"""Does the DDQN trainer learn through the hand-written kernels?  Trains for --steps env steps (DDQN.train: epsilon-greedy
self-play, replay, learner on tron_conv3x3_fwd / _dgrad / _wgrad), then lets the greedy policy play --games games as
player 1 against a uniformly random player 2 and prints the win / loss / draw split next to that of the untrained net
with the same initial weights.  usage: train_sanity.py [--steps 3000] [--envs 4096] [--width 10] [--games 8192]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "deep-q-learning_tron_amd")]
import config  # noqa: F401,E402
import torch  # noqa: E402
import DDQN  # noqa: E402
from tron.vec import VecTron  # noqa: E402


def versus_random(net, width, games, seed):
    env = VecTron(games, width, mode=None, seed=seed, rank=7, obs_format="codes", reward="ddqn")
    S = width + 2
    obs = env.reset()
    wins = torch.zeros(3, dtype=torch.int64, device="cuda")              # draws, player 1, player 2
    finished = torch.zeros(games, dtype=torch.bool, device="cuda")
    for _ in range(4 * width * width):
        a1 = net.infer(obs[:, 0].reshape(games, S, S).contiguous(), codes=True, greedy=True)
        a2 = torch.randint(0, 4, (games,), device="cuda", dtype=torch.int8)
        obs, _, done, winner = env.step(torch.stack([a1, a2], 1).contiguous(), autoreset=True)
        new = done.bool() & ~finished                                     # every env's FIRST game counts
        wins += torch.bincount(winner[new].long(), minlength=3)[:3]
        finished |= done.bool()
        if bool(finished.all()):
            break
    return [int(v) for v in wins.cpu()]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=3000)
    ap.add_argument("--envs", type=int, default=4096)
    ap.add_argument("--width", type=int, default=10)
    ap.add_argument("--games", type=int, default=8192)
    a = ap.parse_args()
    torch.manual_seed(1)
    brain = DDQN.Agent(a.width, 3, buffer_size=1 << 20, batch_size=4096, seed=1, rank=0, make_memory=True)
    before = versus_random(brain.qnetwork_local, a.width, a.games, seed=99)
    out = DDQN.train(n_envs=a.envs, width=a.width, steps=a.steps, batch_size=4096, in_channels=3, log_every=0, brain=brain)
    after = versus_random(brain.qnetwork_local, a.width, a.games, seed=99)
    fmt = lambda w: f"wins {w[1] / sum(w):.3f}  losses {w[2] / sum(w):.3f}  draws {w[0] / sum(w):.3f}"
    print(f"{a.envs} envs x {a.steps} steps ({out['learn_steps']} learn steps of 4096, {out['games']} games, "
          f"{out['env_steps_per_s'] / 1e3:.0f} K env-steps/s, eps {out['epsilon']:.3f})")
    print(f"greedy policy vs uniformly random opponent over {a.games} games, untrained: {fmt(before)}")
    print(f"greedy policy vs uniformly random opponent over {a.games} games, trained:   {fmt(after)}")


if __name__ == "__main__":
    main()
