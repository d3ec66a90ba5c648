#!/usr/bin/env python3
"""Per-kernel timing of the learner's trunk on the weight-stationary design (csrc/tron_conv_ws_train.hip) beside the kernels it
replaces (csrc/tron_conv_f16.hip's chunked forward / input gradient, tron_conv_wgrad*.hip), at the learn batch.
TFLOP/s are f32-equivalent: 2 B S^2 9 cin cout / t; the ceiling is the dense f16 peak / 3 = 833.
usage: trunk_px_bench.py [B] [S]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "deep-q-learning_tron_amd")]
import config, torch  # noqa
from Net import fused, activations
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
S = int(sys.argv[2]) if len(sys.argv) > 2 else 12


def timeit(fn, n=10):
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


def rand_px(C, scale=1.0):
    px = fused.PX16(B, C, S, "cuda")
    # random f16 payload: hi in [-2, 2) * scale / 64, lo small — realistic magnitudes for the clock the chip holds
    n = B * C * S * S
    hi = (torch.randn(n, device="cuda") * scale / 64).to(torch.float16)
    lo = (torch.randn(n, device="cuda") * 0.3).to(torch.float16)
    px.buf.view(torch.float16).reshape(B, 2, -1)[:, 0].copy_(hi.reshape(B, -1))
    px.buf.view(torch.float16).reshape(B, 2, -1)[:, 1].copy_(lo.reshape(B, -1))
    return px


def grad_px(C):
    px = rand_px(C, 128.0)
    g = fused.GradPX(B, C, S, "cuda")
    g.buf = px.buf
    g.info = torch.full((68,), 2.0 ** -7, device="cuda")
    g.info[0], g.info[1] = 2.0 ** 20, 2.0 ** -20
    return g


print(f"B = {B}, {S}x{S}")
print(f"{'layer':22s} {'new us':>9s} {'TF/s':>7s} {'frac':>6s} | {'old us':>9s} {'TF/s':>7s}")
for cin, cout, res in ((32, 32, False), (32, 32, True), (32, 64, False), (64, 64, False), (64, 64, True)):
    fl = 2.0 * B * S * S * 9 * cin * cout
    conv = torch.nn.Conv2d(cin, cout, 3, padding=1).cuda()
    x, r = rand_px(cin), (rand_px(cout) if res else None)
    frag = fused._split_jobs([conv.weight], "tron_conv3x3_ws_split_weights", False)[0]
    t_new = timeit(lambda: fused.conv_ws_train(x, cout, frag, conv.bias, residual=r))
    xf, rf = torch.randn(B, cin, S, S, device="cuda"), (torch.randn(B, cout, S, S, device="cuda") if res else None)
    ws = fused.split_weights([conv])
    t_old = timeit(lambda: fused.conv3x3_raw(xf, conv.weight, conv.bias, rf, act=True, want_pre=True, presplit=ws[0]))
    print(f"fwd  {cin:2d}->{cout:2d}{' +res' if res else '     '}       {t_new:9.1f} {fl / t_new / 1e6:7.0f} {fl / t_new / 1e6 / 833:6.3f} | {t_old:9.1f} {fl / t_old / 1e6:7.0f}")
    del xf, rf
    # input gradient (+ residual gradient for the layers whose input feeds a skip connection) with the activation backward
    g, zb, ex = grad_px(cout), rand_px(cin), (grad_px(cin) if res else None)
    rot, wn = fused._split_jobs([conv.weight], "tron_conv3x3_ws_split_weights_bwd", True)
    t_new = timeit(lambda: fused.conv_ws_dgrad(g, cin, rot[0], wn[0:1], zb, extra=ex))
    gf, zf = torch.randn(B, cout, S, S, device="cuda") * 1e-6, torch.randn(B, cin, S, S, device="cuda")
    exf = torch.randn(B, cin, S, S, device="cuda") * 1e-6 if res else None
    am = torch.full((cout,), 4e-6, device="cuda")
    t_old = timeit(lambda: fused.conv3x3_dgrad_mish(gf, conv.weight, am, zf, extra=exf))
    print(f"dgrad {cout:2d}->{cin:2d}{' +ext' if res else '     '}      {t_new:9.1f} {fl / t_new / 1e6:7.0f} {fl / t_new / 1e6 / 833:6.3f} | {t_old:9.1f} {fl / t_old / 1e6:7.0f}")
    if not res:
        t_new = timeit(lambda: fused.conv3x3_wgrad_px(x, g))
        xf = torch.randn(B, cin, S, S, device="cuda")
        t_old = timeit(lambda: fused.conv3x3_wgrad(xf, gf, am))
        print(f"wgrad {cin:2d}->{cout:2d}           {t_new:9.1f} {fl / t_new / 1e6:7.0f} {fl / t_new / 1e6 / 833:6.3f} | {t_old:9.1f} {fl / t_old / 1e6:7.0f}")
    del g, zb, ex, gf, zf, exf
    torch.cuda.empty_cache()
