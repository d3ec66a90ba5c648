#!/usr/bin/env python3
"""Host side of Agent.learn(): wall time per step with the device kept busy (back to back) and with a synchronise after every step's
launches (host time alone = launches issued before the device finished), then cProfile's top entries.  usage: learn_cpu_profile.py [width] [batch]"""
import os, sys, time, cProfile, pstats, io
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "deep-q-learning_tron_amd")]
import config, torch  # noqa
import DDQN
W = int(sys.argv[1]) if len(sys.argv) > 1 else 10
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
S = W + 2
brain = DDQN.Agent(W, 3, buffer_size=1 << 16, batch_size=batch, seed=1)
vals = torch.tensor([1, -1, -2, -3, 10, -10], dtype=torch.int8, device="cuda")
c = vals[torch.randint(0, 6, (1 << 15, S, S), device="cuda")]
z = torch.zeros(1 << 15, device="cuda")
brain.memory.add_batch(c, z.to(torch.int8), z, c.flip(0), z.to(torch.int8))
for _ in range(5):
    brain.learn(brain.memory.sample_codes(), DDQN.GAMMA)
torch.cuda.synchronize()
n = 30
t0 = time.perf_counter()
for _ in range(n):
    brain.learn(brain.memory.sample_codes(), DDQN.GAMMA)
t_issue = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print(f"{W}x{W} batch {batch}: {t_all / n * 1e3:.2f} ms per learn step; the host had issued all {n} steps after {t_issue / n * 1e3:.2f} ms per step")
host = 0.0
for _ in range(n):
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    brain.learn(brain.memory.sample_codes(), DDQN.GAMMA)
    host += time.perf_counter() - t1
torch.cuda.synchronize()
print(f"host time per step when the device is idle at its start (pure issue cost): {host / n * 1e3:.2f} ms")
pr = cProfile.Profile()
pr.enable()
for _ in range(20):
    brain.learn(brain.memory.sample_codes(), DDQN.GAMMA)
pr.disable()
torch.cuda.synchronize()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(28)
print(s.getvalue()[:6000])
