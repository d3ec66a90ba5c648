#!/usr/bin/env python3
"""conv1's weight gradient kernel alone (tron_conv1_wgrad_px16) at the learn batch; usage: c1w_bench.py [B]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "deep-q-learning_tron_amd")]
import config, torch  # noqa
from Net import fused
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
for S in (12, 26):
    codes = torch.tensor([1, -1, -2, -3, 10, -10], dtype=torch.int8, device="cuda")[torch.randint(0, 6, (B, S, S), device="cuda")]
    g = fused.GradPX(B, 32, S, "cuda")
    g.buf.view(torch.float16).copy_((torch.randn(g.buf.numel() // 2, device="cuda") * 0.01).to(torch.float16))
    g.info = torch.ones(68, device="cuda")
    for _ in range(3):
        fused.conv1_wgrad_px(codes, g, 3, 0.0)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    torch.cuda.synchronize()
    ev[0].record()
    for _ in range(20):
        fused.conv1_wgrad_px(codes, g, 3, 0.0)
    ev[1].record()
    torch.cuda.synchronize()
    t = ev[0].elapsed_time(ev[1]) / 20 * 1e3
    print(f"{B} x {S}x{S}: {t:7.1f} us per call (kernel + finish), gradient image {g.buf.numel() / 1e6:.0f} MB -> {g.buf.numel() / t / 1e6:.2f} TB/s")
