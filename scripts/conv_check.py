#!/usr/bin/env python3
"""GPU check + timing of the fused 3x3 convolution (csrc/tron_conv.hip, tron_conv_f16.hip) against torch (MIOpen).
Usage: python scripts/conv_check.py [batch12 batch26 [maths]]      maths: f32,f16x3"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "deep-q-learning_tron_amd")]
import config  # noqa: F401,E402  (MIOpen env defaults)
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402
from Net import fused  # noqa: E402
from tron.vec import pop_up_planes  # noqa: E402

dev = "cuda"
B12 = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
B26 = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
MATHS = sys.argv[3].split(",") if len(sys.argv) > 3 else ["f32", "f16x3"]


def timeit(fn, n=10):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


def check_layer(math, S, B, cin, cout):
    conv = torch.nn.Conv2d(cin, cout, 3, padding=1).to(dev)
    x = torch.randn(B, cin, S, S, device=dev)
    res = torch.randn(B, cout, S, S, device=dev)
    got, pre = fused.conv3x3(x, conv, residual=res, want_pre=True, math=math)
    ref64 = F.conv2d(x.double(), conv.weight.double(), conv.bias.double(), padding=1) + res.double()
    ref32 = F.conv2d(x, conv.weight, conv.bias, padding=1) + res
    e_pre = (pre.double() - ref64).abs().max().item()
    e_mi = (ref32.double() - ref64).abs().max().item()
    e_out = (got.double() - F.mish(ref64)).abs().max().item()
    rms = (pre.double() - ref64).pow(2).mean().sqrt().item()
    line = f"[{math}] S={S} B={B} {cin}->{cout}: |hip-f64| max {e_pre:.2e} rms {rms:.2e} (MIOpen f32 max {e_mi:.2e}) out {e_out:.2e}"
    if B >= 1024:
        t_h = timeit(lambda: fused.conv3x3(x, conv, residual=res, math=math))
        t_m = timeit(lambda: F.mish(F.conv2d(x, conv.weight, conv.bias, padding=1) + res))
        fl = 2 * B * S * S * cout * cin * 9
        line += f" | hip {t_h * 1e3:.3f} ms = {fl / t_h / 1e12:.1f} TF/s, torch {t_m * 1e3:.3f} ms = {fl / t_m / 1e12:.1f} TF/s"
    print(line, flush=True)
    assert e_pre < 5e-5 and e_out < 5e-5


def check_conv1(math, S, B, cin):
    conv = torch.nn.Conv2d(cin, 32, 3, padding=1).to(dev)
    vals = torch.tensor([1, -1, -2, -3, 10, -10], dtype=torch.int8, device=dev)
    codes = vals[torch.randint(0, 6, (B, S, S), device=dev)]
    planes = pop_up_planes(codes)
    if cin == 4:
        planes = torch.cat([planes, torch.full((B, 1, S, S), 5.0, device=dev)], 1)
    got = fused.conv3x3(codes, conv, codes=True, plane4=5.0, math=math)
    got_p = fused.conv3x3(planes, conv, math=math)
    ref = F.mish(F.conv2d(planes.double(), conv.weight.double(), conv.bias.double(), padding=1))
    e = (got.double() - ref).abs().max().item()
    print(f"[{math}] S={S} B={B} conv1 from codes, cin={cin}: err {e:.2e}, codes == planes path: {torch.equal(got, got_p)}", flush=True)
    assert e < 5e-5 and torch.equal(got, got_p)


torch.manual_seed(0)
for math in MATHS:
    for S, B in ((12, B12), (26, B26), (12, 5), (26, 3)):
        for cin, cout in ((32, 32), (32, 64), (64, 64)):
            check_layer(math, S, B, cin, cout)
        for cin in (3, 4):
            check_conv1(math, S, B, cin)
print("conv check ok")
