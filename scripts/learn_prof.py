#!/usr/bin/env python3
"""Agent.learn() back to back on a filled ring, for rocprofv3 --kernel-trace; usage: learn_prof.py [width] [batch] [iters]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "deep-q-learning_tron_amd")]
import config, torch  # noqa
import DDQN
W = int(sys.argv[1]) if len(sys.argv) > 1 else 10
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
n = int(sys.argv[3]) if len(sys.argv) > 3 else 10
S = W + 2
brain = DDQN.Agent(W, 3, buffer_size=1 << 16, batch_size=batch, seed=1)
vals = torch.tensor([1, -1, -2, -3, 10, -10], dtype=torch.int8, device="cuda")
c = vals[torch.randint(0, 6, (1 << 15, S, S), device="cuda")]
z = torch.zeros(1 << 15, device="cuda")
brain.memory.add_batch(c, z.to(torch.int8), z, c.flip(0), z.to(torch.int8))
for _ in range(3):
    brain.learn(brain.memory.sample_codes(), DDQN.GAMMA)
torch.cuda.synchronize()
import time
t0 = time.perf_counter()
for _ in range(n):
    brain.learn(brain.memory.sample_codes(), DDQN.GAMMA)
torch.cuda.synchronize()
print(f"learn step {W}x{W} batch {batch}: {(time.perf_counter() - t0) / n * 1e3:.2f} ms")
