#!/usr/bin/env python3
"""Time tron_conv3x3_fwd against the number of workgroups: 1 per CU, 2 per CU, ... (co-residency / overlap check)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "deep-q-learning_tron_amd")]
import config, torch  # noqa
from Net import fused
conv = torch.nn.Conv2d(32, 32, 3, padding=1).cuda()
for B in (256, 1024, 2048, 3072, 4096, 8192, 16384):
    x = torch.randn(B, 32, 12, 12, device="cuda")
    r = torch.randn(B, 32, 12, 12, device="cuda")
    for _ in range(3):
        fused.conv3x3(x, conv, residual=r)
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(20):
        fused.conv3x3(x, conv, residual=r)
    ev[1].record()
    torch.cuda.synchronize()
    t = ev[0].elapsed_time(ev[1]) / 20
    print(f"B={B:6d} WGs={B // 4:5d} ({B / 4 / 256:.1f}/CU): {t * 1e3:8.1f} us  {2 * B * 144 * 32 * 288 / t / 1e9:.1f} TF/s", flush=True)
