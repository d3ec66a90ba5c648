#!/usr/bin/env python3
"""Time the weight-gradient kernel (tron_conv3x3_wgrad) per layer shape against aten.convolution_backward (MIOpen);
usage: wgrad_bench.py [B] [side]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "deep-q-learning_tron_amd")]
import config, torch  # noqa
from Net import fused

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
S = int(sys.argv[2]) if len(sys.argv) > 2 else 12


def timed(fn, n=10):
    for _ in range(3):
        fn()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(n):
        fn()
    ev[1].record()
    torch.cuda.synchronize()
    return ev[0].elapsed_time(ev[1]) / n * 1e3


for cin, cout in (((4, 32),) if S == 12 else ()) + ((32, 32), (32, 64), (64, 64)):
    x = torch.randn(B, cin, S, S, device="cuda")
    gp = torch.randn(B, cout, S, S, device="cuda") * 1e-4
    w = torch.randn(cout, cin, 3, 3, device="cuda")
    absmax = gp.abs().reshape(64, -1).amax(1).contiguous()
    fl = 2 * B * S * S * 9 * cin * cout
    t = timed(lambda: fused.conv3x3_wgrad(x, gp, absmax))
    t0 = timed(lambda: fused.conv3x3_wgrad(x, gp, None))
    ta = timed(lambda: torch.ops.aten.convolution_backward(gp, x, w, None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1, [False, True, False]))
    line = f"wgrad {cin:2d}->{cout:2d} B={B}: {t:7.1f} us ({fl / t / 1e6:6.1f} TF/s f32-equivalent; {t0:7.1f} us with the max pre-pass)  aten {ta:7.1f} us"
    if cin >= 32:
        td = timed(lambda: fused.conv3x3_dgrad(gp, w, absmax))
        line += f"   dgrad {td:7.1f} us"
    print(line, flush=True)
