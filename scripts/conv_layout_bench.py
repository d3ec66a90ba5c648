#!/usr/bin/env python3
"""Per-layer forward+backward time of the DQN CNN's convolutions, NCHW vs channels_last (MIOpen)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "deep-q-learning_tron_amd"))
import config  # noqa: F401,E402  (MIOpen env defaults)
import torch  # noqa: E402
import torch.nn as nn  # noqa: E402

B, S = int(sys.argv[1]) if len(sys.argv) > 1 else 4096, int(sys.argv[2]) if len(sys.argv) > 2 else 12
layers = [("conv1 3->32 k3", 3, 32, 3, 1, 1, S), ("conv2 32->32 k3", 32, 32, 3, 1, 1, S), ("conv4 32->64 k3", 32, 64, 3, 1, 1, S),
          ("conv5 64->64 k3", 64, 64, 3, 1, 1, S), ("conv7 64->64 k7 s2", 64, 64, 7, 3, 2, (S + 1) // 2)]
for name, ci, co, k, p, st, side in layers:
    row = []
    for cl in (False, True):
        conv = nn.Conv2d(ci, co, k, padding=p, stride=st).cuda()
        x = torch.randn(B, ci, side, side, device="cuda", requires_grad=True)
        if cl:
            conv = conv.to(memory_format=torch.channels_last)
            x = x.detach().contiguous(memory_format=torch.channels_last).requires_grad_(True)

        def step():
            y = conv(x)
            y.backward(torch.ones_like(y))
        for _ in range(3):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            step()
        torch.cuda.synchronize()
        row.append((time.perf_counter() - t0) / 10 * 1e3)
    print(f"{name:22s} NCHW {row[0]:8.3f} ms   channels_last {row[1]:8.3f} ms", flush=True)
