#!/usr/bin/env python3
"""One conv layer call repeated (for rocprofv3 --pmc passes): conv_pmc.py [B cin cout S math in_fmt]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "deep-q-learning_tron_amd")]
import config, torch  # noqa
from Net import fused
B, cin, cout, S = (int(v) for v in (sys.argv[1:5] + ["8192", "64", "64", "12"][len(sys.argv) - 1:])[:4])
math = sys.argv[5] if len(sys.argv) > 5 else "f16x3"
conv = torch.nn.Conv2d(cin, cout, 3, padding=1).cuda()
x = torch.randn(B, cin, S, S, device="cuda")
r = torch.randn(B, cout, S, S, device="cuda")
for _ in range(6):
    fused.conv3x3(x, conv, residual=r, math=math)
torch.cuda.synchronize()
