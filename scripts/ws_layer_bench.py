#!/usr/bin/env python3
"""Per-layer timing of the weight-stationary chain (csrc/tron_conv_ws.hip) against the chunked split-f16 kernel
(csrc/tron_conv_f16.hip) on the same shapes; usage: ws_layer_bench.py [B] [S]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "deep-q-learning_tron_amd")]
import config, torch  # noqa
from Net import fused
from Net.DQNNet import Net
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
S = int(sys.argv[2]) if len(sys.argv) > 2 else 12
net = Net(3, S - 2).cuda()
vals = torch.tensor([1, -1, -2, -3, 10, -10], dtype=torch.int8, device="cuda")
codes = vals[torch.randint(0, 6, (B, S, S), device="cuda")]


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    torch.cuda.synchronize()
    ev[0].record()
    for _ in range(n):
        fn()
    ev[1].record()
    torch.cuda.synchronize()
    return ev[0].elapsed_time(ev[1]) / n * 1e3


w = fused.ws_split_weights([net.conv2, net.conv3, net.conv4, net.conv5, net.conv6])
a = fused.conv1_px16(codes, net.conv1)
b = fused.conv_ws(a, net.conv2, w[0])
c = fused.conv_ws(b, net.conv3, w[1], residual=a)
d = fused.conv_ws(c, net.conv4, w[2])
e = fused.conv_ws(d, net.conv5, w[3])
L = fused.nat.lib()
nat = fused.nat


def raw(x, conv, wf, res, out, o32=None):
    def f():
        nat.check(L.tron_conv3x3_ws_fwd(nat.ptr(x.buf), nat.ptr(wf), nat.ptr(conv.bias.detach()), nat.ptr(None if res is None else res.buf),
                                        nat.ptr(None if out is None else out.buf), nat.ptr(o32), None, B, conv.in_channels, conv.out_channels, S, 1, nat.stream_ptr()))
    return f


o32 = torch.empty(B, 64, S, S, device="cuda")
rows = [("conv1 codes->px16", lambda: fused.conv1_px16(codes, net.conv1), 3, 32),
        ("conv2 32->32", raw(a, net.conv2, w[0], None, b), 32, 32),
        ("conv3 32->32 +res", raw(b, net.conv3, w[1], a, c), 32, 32),
        ("conv4 32->64", raw(c, net.conv4, w[2], None, d), 32, 64),
        ("conv5 64->64", raw(d, net.conv5, w[3], None, e), 64, 64),
        ("conv6 64->64 +res", raw(e, net.conv6, w[4], d, e.__class__(B, 64, S, e.buf.device)), 64, 64),
        ("conv6 64->64 +res -> f32", raw(e, net.conv6, w[4], d, None, o32), 64, 64),
        ("split weights (5 layers)", lambda: fused.ws_split_weights([net.conv2, net.conv3, net.conv4, net.conv5, net.conv6]), 0, 0)]
only = sys.argv[3] if len(sys.argv) > 3 else None
if only:
    rows = [r for r in rows if r[0].startswith(only)]
for name, fn, ci, co in rows:
    t = timeit(fn)
    fl = 2.0 * B * S * S * 9 * ci * co
    print(f"{name:28s} {t:8.1f} us  {fl / t / 1e6:7.1f} TF/s f32-eq", flush=True)
for math in (() if only else ("f16x3",)):
    for ws in (True, False):
        fused.use_ws = ws
        t = timeit(lambda: fused.trunk(net, codes, codes=True, math=math), 10)
        fl = 2 * B * S * S * 9 * (3 * 32 + 2 * 32 * 32 + 32 * 64 + 2 * 64 * 64)
        print(f"trunk ws={ws}: {t:.1f} us = {fl / t / 1e6:.1f} TF/s f32-equivalent", flush=True)
        t = timeit(lambda: net.infer(codes, codes=True, greedy=True), 10)
        print(f"Net.infer ws={ws}: {t:.1f} us", flush=True)
if not only:
    fused.use_ws = True
    for sl in (512, 1024, 2048, 4096, B):
        if sl > B:
            continue
        def sliced():
            outs = []
            w = fused.ws_split_weights([net.conv2, net.conv3, net.conv4, net.conv5, net.conv6])
            for i in range(0, B, sl):
                c_ = codes[i:i + sl]
                a = fused.conv1_px16(c_, net.conv1)
                b = fused.conv_ws(a, net.conv2, w[0])
                c = fused.conv_ws(b, net.conv3, w[1], residual=a)
                d = fused.conv_ws(c, net.conv4, w[2])
                e = fused.conv_ws(d, net.conv5, w[3])
                outs.append(fused.conv_ws(e, net.conv6, w[4], residual=d))
            return outs
        t = timeit(sliced, 10)
        print(f"trunk (px16 out) in slices of {sl}: {t:.1f} us", flush=True)
