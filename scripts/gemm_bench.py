#!/usr/bin/env python3
"""tron_gemm_f16x3 (fused.gemm_f16x3) at the head's shapes: HIP-event time per call.  usage: gemm_bench.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "deep-q-learning_tron_amd")]
import config, torch  # noqa
from Net import fused


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


# (what, M, N, K, a_transposed, b_transposed)
shapes = [("conv7 dense fwd, B 4096", 4096, 576, 2304, False, False), ("conv7 dense fwd, B 8192", 8192, 576, 2304, False, False),
          ("conv7 dense dX, B 4096", 4096, 2304, 576, False, True), ("conv7 dense dW, B 4096", 576, 2304, 4096, True, True),
          ("fc1 fwd, B 4096", 4096, 256, 576, False, False), ("conv7 dense fwd, B 1024", 1024, 576, 2304, False, False),
          ("conv7 dense fwd, B 65536", 65536, 576, 2304, False, False)]
for what, M, N, K, at, bt in shapes:
    a = torch.randn((K, M) if at else (M, K), device="cuda")
    b = torch.randn((K, N) if bt else (N, K), device="cuda")
    ref = (a.t() if at else a).double() @ (b if bt else b.t()).double()
    out = fused.gemm_f16x3(a, b, a_transposed=at, b_transposed=bt)
    err = (out.double() - ref).abs().max().item() / ref.abs().max().item()
    t = timed(lambda: fused.gemm_f16x3(a, b, a_transposed=at, b_transposed=bt))
    print(f"{what:28s} M {M:6d} N {N:5d} K {K:5d}: {t:8.1f} us  ({2 * M * N * K / t * 1e-6:6.1f} TFLOP/s f32-equivalent, operand splits included)  rel err {err:.1e}")
