#!/usr/bin/env python3
"""K-FAC factor kernels (csrc/tron_kfac.hip) alone: ms and TFLOP/s (full-product-equivalent 2 d^2 rows) per layer shape of the
ACKTR nets at BASELINE config 5, one micro-batch of 8 192 samples; usage: gram_bench.py [batch]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "deep-q-learning_tron_amd")]
import config, torch  # noqa
import torch.nn as nn
from Net import kfac
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
shapes = [("conv1 3->32, 34x34", 3, 34, 3, 1, 1), ("conv2/3/4 (cin 32), 34x34", 32, 34, 3, 1, 1), ("conv5/6 (cin 64), 34x34", 64, 34, 3, 1, 1),
          ("conv7 7x7/2 on 17x17", 64, 17, 7, 3, 2), ("G of a 64-channel conv, 34x34 (1x1)", 64, 34, 1, 0, 1)]
for name, C, S, k, pad, stride in shapes:
    x = torch.randn(B, C, S, S, device="cuda")
    conv = nn.Conv2d(C, 8, k, padding=pad, stride=stride)
    for _ in range(2):
        kfac._gram_hip(x, conv, 1.0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 5
    for _ in range(n):
        kfac._gram_hip(x, conv, 1.0)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    oh = (S + 2 * pad - k) // stride + 1
    d = C * k * k
    print(f"{name:40s} d={d:5d} rows={B * oh * oh:9d}: {dt * 1e3:8.2f} ms  {2.0 * d * d * B * oh * oh / dt / 1e12:7.1f} TFLOP/s full-product-equivalent")
a = torch.randn(B, 5184, device="cuda")
for _ in range(2):
    kfac._gram_hip(a, nn.Linear(5184, 4), 1.0)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    kfac._gram_hip(a, nn.Linear(5184, 4), 1.0)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 5
print(f"{'fc1 input (64 x 9 x 9)':40s} d= 5184 rows={B:9d}: {dt * 1e3:8.2f} ms  {2.0 * 5184 * 5184 * B / dt / 1e12:7.1f} TFLOP/s full-product-equivalent")
