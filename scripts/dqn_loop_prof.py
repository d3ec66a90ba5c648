#!/usr/bin/env python3
"""A few iterations of the batched DDQN trainer (DDQN.train) for rocprofv3 --kernel-trace; usage: dqn_loop_prof.py [envs] [width] [steps]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "deep-q-learning_tron_amd")]
import config, torch  # noqa
import DDQN
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
w = int(sys.argv[2]) if len(sys.argv) > 2 else 10
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 12
brain = None
for rep in range(2):                       # the second call is the steady state
    out = DDQN.train(n_envs=n, width=w, steps=steps, learn_every=2, batch_size=4096, capacity=1 << 18, log_every=0, seed=3)
torch.cuda.synchronize()
print(out["learned_transitions_per_s"])
