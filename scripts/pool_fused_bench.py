#!/usr/bin/env python3
"""conv6 (64 -> 64, 12x12, + residual) as tron_conv3x3_ws_fwd against tron_conv3x3_ws_fwd_pool12 (the pooling inside), and the head
from the PX16 image against the head from the pooled rows: HIP-event times.  usage: pool_fused_bench.py [batch ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "deep-q-learning_tron_amd")]
import config, torch  # noqa
from Net import fused
from Net.DQNNet import Net


def timed(fn, n=30):
    for _ in range(5):
        fn()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    torch.cuda.synchronize()
    ev[0].record()
    for _ in range(n):
        fn()
    ev[1].record()
    torch.cuda.synchronize()
    return ev[0].elapsed_time(ev[1]) / n * 1e3


for B in [int(a) for a in sys.argv[1:]] or [4096, 8192]:
    net = Net(3, 10).cuda().eval()
    x, r = fused.PX16(B, 64, 12, "cuda"), fused.PX16(B, 64, 12, "cuda")
    for t in (x, r):
        t.buf.view(torch.float16).copy_(torch.randn(t.buf.numel() // 2, device="cuda") * 0.02)
    w = fused.ws_split_weights([net.conv6])[0]
    with torch.no_grad():
        y = fused.conv_ws(x, net.conv6, w, residual=r)
        p = fused.conv_ws_pool12(x, net.conv6, w, r)
        a = timed(lambda: fused.conv_ws(x, net.conv6, w, residual=r))
        b = timed(lambda: fused.conv_ws_pool12(x, net.conv6, w, r))
        c = timed(lambda: fused.head(net, y))
        d = timed(lambda: fused.head(net, p))
    print(f"{B:6d} x 12x12: conv6 {a:7.1f} us, conv6 + pooling in one launch {b:7.1f} us; head from PX16 {c:7.1f} us, from pooled rows {d:7.1f} us;"
          f" together {a + c:7.1f} -> {b + d:7.1f} us")
