#!/usr/bin/env python3
"""Net.infer(codes) a few times, for rocprofv3 --kernel-trace --stats; usage: infer_prof.py [B] [S] [iters]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "deep-q-learning_tron_amd")]
import config, torch  # noqa
from Net.DQNNet import Net
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
S = int(sys.argv[2]) if len(sys.argv) > 2 else 12
n = int(sys.argv[3]) if len(sys.argv) > 3 else 20
net = Net(3, S - 2).cuda()
vals = torch.tensor([1, -1, -2, -3, 10, -10], dtype=torch.int8, device="cuda")
codes = vals[torch.randint(0, 6, (B, S, S), device="cuda")]
print("path:", net.infer_path(codes, codes=True))
for _ in range(n):
    g = net.infer(codes, codes=True, greedy=True)
torch.cuda.synchronize()
