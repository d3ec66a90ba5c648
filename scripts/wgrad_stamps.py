#!/usr/bin/env python3
"""Diagnostic (-DTRON_WG_STAMPS build of the library, TRON_HIP_LIB pointing at it): per-workgroup shader-clock stamps
of the weight-gradient kernel — prologue, loop, the part of the loop spent between the two barriers (LDS stores)."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "deep-q-learning_tron_amd")]
import config, torch  # noqa
import numpy as np
from Net import fused
from tron import _native as nat
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
for cin, cout in ((32, 32), (32, 64), (64, 64)):
    x = torch.randn(B, cin, 12, 12, device="cuda")
    gp = torch.randn(B, cout, 12, 12, device="cuda") * 1e-4
    absmax = gp.abs().reshape(64, -1).amax(1).contiguous()
    for _ in range(300):                 # warm the clocks up; the last launch's stamps are read
        fused.conv3x3_wgrad(x, gp, absmax)
    torch.cuda.synchronize()
    buf = (C.c_ulonglong * (256 * 8))()
    assert nat.lib().tron_wgrad_stamps(buf) == 0
    s = np.frombuffer(buf, dtype=np.uint64).reshape(256, 8).astype(np.float64)
    nwg = min(256, B * (cout // 32))
    s = s[:nwg]
    pro, loop, stage = s[:, 1] - s[:, 0], s[:, 3] - s[:, 1], s[:, 5]
    rt = (s[:, 4] - s[:, 2]) * 10.0     # ns
    rounds = B / (256 // (cout // 32))
    cit = cin // 16
    ideal = rounds * (6 // (4 // cit)) * 54 * 16
    print(f"{cin}->{cout}: prologue {np.median(pro):.0f} cyc; loop {np.median(loop):.0f} cyc = {np.median(rt) / 1e3:.1f} us "
          f"(clock {np.median(loop / rt):.2f} GHz); between the barriers {np.median(stage):.0f} cyc; "
          f"MFMA section {np.median(loop - stage):.0f} cyc vs {ideal:.0f} ideal; span {(s[:, 4].max() - s[:, 2].min()) * 10 / 1e3:.1f} us")
