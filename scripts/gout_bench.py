#!/usr/bin/env python3
"""tron_px16_grad_from_pooled (the gradient chain's entry: pooling backward x mish'(z6) -> gradient image + bias sums) alone: HIP-event
time per call.  usage: gout_bench.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "deep-q-learning_tron_amd")]
import config, torch  # noqa
from Net import fused
from tron import _native as nat
L = nat.lib()
for S, B in ((12, 4096), (26, 4096)):
    C, PS = 64, S // 2
    z = fused.PX16(B, C, S, "cuda")
    z.buf.view(torch.float16).copy_(torch.randn(z.buf.numel() // 2, device="cuda") * 0.02)
    g = torch.randn(B, PS, PS, C, device="cuda") if S == 26 else torch.randn(B, C, PS, PS, device="cuda")
    sc4 = torch.zeros(4, device="cuda")
    out = fused.GradPX(B, C, S, "cuda")
    gb = torch.empty(C, device="cuda")
    ws = torch.empty(int(L.tron_px16_grad_workspace(B, C)), dtype=torch.uint8, device="cuda")
    nat.check(L.tron_absmax_pow2(nat.ptr(g), g.numel(), 15, nat.ptr(sc4), nat.stream_ptr()))
    fn = lambda: nat.check(L.tron_px16_grad_from_pooled(nat.ptr(g), int(S == 26), nat.ptr(z.buf), nat.ptr(sc4), B, C, S, nat.ptr(out.buf),
                                                        nat.ptr(out.info), nat.ptr(gb), nat.ptr(ws), nat.stream_ptr()))
    for _ in range(5):
        fn()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    torch.cuda.synchronize()
    ev[0].record()
    for _ in range(30):
        fn()
    ev[1].record()
    torch.cuda.synchronize()
    t = ev[0].elapsed_time(ev[1]) / 30 * 1e3
    nbytes = B * C * (S * S * 8 + PS * PS * 4)
    print(f"chain entry {B} x {C} x {S}x{S}: {t:7.1f} us  ({nbytes / t * 1e-6:.2f} TB/s of z image + gradient image + pooled gradient)")
