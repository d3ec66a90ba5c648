#!/usr/bin/env python3
"""Agent.learn() wall time at 10x10 with conv6 + pooling as one launch in (a) the target forwards and the training forward, (b) the
target forwards only, (c) neither — interleaved repeats in one process.  usage: learn_time_ab.py [batch] [iters]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "deep-q-learning_tron_amd")]
import config, torch  # noqa
import DDQN
from Net import fused
batch = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
W, S = 10, 12
brain = DDQN.Agent(W, 3, buffer_size=1 << 16, batch_size=batch, seed=1)
vals = torch.tensor([1, -1, -2, -3, 10, -10], dtype=torch.int8, device="cuda")
c = vals[torch.randint(0, 6, (1 << 15, S, S), device="cuda")]
z = torch.zeros(1 << 15, device="cuda")
brain.memory.add_batch(c, z.to(torch.int8), z, c.flip(0), z.to(torch.int8))


def timed():
    for _ in range(5):
        brain.learn(brain.memory.sample_codes(), DDQN.GAMMA)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        brain.learn(brain.memory.sample_codes(), DDQN.GAMMA)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


res = {"both": [], "targets only": [], "neither": []}
for _ in range(4):
    for name, (i, t) in (("both", (True, True)), ("targets only", (True, False)), ("neither", (False, False))):
        fused.use_pool_fused, fused.use_pool_fused_train = i, t
        res[name].append(timed())
for k, v in res.items():
    print(f"learn step 10x10 batch {batch}, one-launch conv6 + pooling in {k:13s}: {min(v):.3f} ms   [{' '.join(f'{x:.3f}' for x in v)}]")
