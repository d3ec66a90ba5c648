#!/usr/bin/env python3
"""Diagnostic (-DTRON_CONV_STAMPS build): per-workgroup shader-clock / real-time stamps of the conv kernel."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "deep-q-learning_tron_amd")]
import config, torch  # noqa
import numpy as np
from tron import _native as nat
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
cin, cout, S = int(os.environ.get("CIN", 32)), int(os.environ.get("COUT", 32)), 12
MATH = int(os.environ.get("MATH", 0))
IN_FMT = int(os.environ.get("IN_FMT", 0))      # 2: the input is a split-f16 image (timing only: random halves)
OUT_S16 = int(os.environ.get("OUT_S16", 0))
PER = 2 if MATH else 4
conv = torch.nn.Conv2d(cin, cout, 3, padding=1).cuda()
x = torch.randn(B, cin, S, S, device="cuda"); r = torch.randn(B, cout, S, S, device="cuda")
out = torch.empty(B, cout, S, S, device="cuda")
x16 = (torch.randn(B * cin * S * S * 2, device="cuda") * 0.1).to(torch.float16)
o16 = torch.empty(B * cout * S * S * 4, dtype=torch.uint8, device="cuda")
blocks = B // PER * (2 if (cout == 64 and not MATH) else 1)
st = torch.zeros(blocks, 6, dtype=torch.int64, device="cuda")
L = nat.lib()
ws = torch.empty(int(L.tron_conv3x3_workspace(cin, cout)), dtype=torch.uint8, device="cuda")
for rep in range(30):          # warm the clocks up: the stamps of the last launch are read
    nat.check(L.tron_conv3x3_fwd(nat.ptr(x16 if IN_FMT == 2 else x), IN_FMT, nat.ptr(conv.weight.detach()), nat.ptr(conv.bias.detach()), nat.ptr(r), nat.ptr(out),
                                 nat.ptr(st), B, cin, cout, S, 0.0, 1, MATH, nat.ptr(ws), nat.ptr(o16) if OUT_S16 else None, nat.stream_ptr()))
torch.cuda.synchronize()
s = st.cpu().numpy().astype(np.float64)
if MATH:      # f16 kernel stamps: start, end of prologue, end of main loop
    pro_rt = (s[:, 3] - s[:, 1]) * 10.0
    main_cyc, main_rt = s[:, 4] - s[:, 2], (s[:, 5] - s[:, 3]) * 10.0
    print(f"prologue median {np.median(pro_rt) / 1e3:.2f} us; main loop median {np.median(main_rt) / 1e3:.2f} us = {np.median(main_cyc):.0f} cycles; "
          f"ideal MFMA cycles per SIMD {cin // 16 * 5 * 9 * (cout // 32) * 3 * 16}; clock {np.median(main_cyc / main_rt):.3f} GHz; "
          f"kernel span {(s[:, 5].max() - s[:, 1].min()) * 10 / 1e3:.1f} us for {blocks} blocks")
    sys.exit(0)
main_cyc, main_rt = s[:, 2] - s[:, 0], (s[:, 3] - s[:, 1]) * 10.0          # realtime ticks are 10 ns
epi_cyc, epi_rt = s[:, 4] - s[:, 2], (s[:, 5] - s[:, 3]) * 10.0
clk = main_cyc / main_rt           # cycles per ns = GHz
t0 = s[:, 1].min()
print(f"B={B} blocks={blocks}: main phase median {np.median(main_rt) / 1e3:.2f} us, {np.median(main_cyc):.0f} cycles, clock {np.median(clk):.3f} GHz "
      f"(p10 {np.percentile(clk, 10):.3f}, p90 {np.percentile(clk, 90):.3f}); epilogue median {np.median(epi_rt) / 1e3:.2f} us {np.median(epi_cyc):.0f} cycles; "
      f"kernel span {(s[:, 5].max() - t0) * 10 / 1e3:.1f} us; ideal MFMA cycles {cin // 8 * 18 * 18 * 32}")
starts = np.sort((s[:, 1] - t0) * 10 / 1e3)
print("start times us (sorted) at block 0,255,256,511,512,1023:", [round(float(starts[min(i, blocks - 1)]), 1) for i in (0, 255, 256, 511, 512, 1023)])
