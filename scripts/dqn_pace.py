#!/usr/bin/env python3
"""Wall time of each block of 20 env-steps of the batched DDQN trainer (is the pace steady?)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "deep-q-learning_tron_amd"))
import torch  # noqa: E402
import DDQN  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
for block in range(int(sys.argv[2]) if len(sys.argv) > 2 else 12):
    t0 = time.perf_counter()
    out = DDQN.train(n_envs=n, width=10, steps=20, learn_every=2, batch_size=4096, capacity=1 << 20, log_every=0,
                     seed=block)
    torch.cuda.synchronize()
    print(f"block {block}: {time.perf_counter() - t0:6.2f} s   {out['learned_transitions_per_s'] / 1e3:7.1f} K transitions/s",
          flush=True)
