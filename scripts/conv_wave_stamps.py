#!/usr/bin/env python3
"""Diagnostic (-DTRON_CONV_WAVE_STAMPS build, TRON_HIP_LIB pointing at it): per-wave cycles of the split-f16 conv kernel —
chunk loop, the part of it spent waiting at the chunk barriers, epilogue — summed over a workgroup's image groups."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "deep-q-learning_tron_amd")]
import config, torch  # noqa
import numpy as np
from Net import fused
from tron import _native as nat
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
cin, cout = int(os.environ.get("CIN", 64)), int(os.environ.get("COUT", 64))
conv = torch.nn.Conv2d(cin, cout, 3, padding=1).cuda()
x = torch.randn(B, cin, 12, 12, device="cuda")
_, s16 = fused.conv3x3(x, torch.nn.Conv2d(cin, cin, 3, padding=1).cuda(), want_split=True)
for _ in range(200):
    fused.conv3x3(s16, conv, want_f32=False, want_split=True)
torch.cuda.synchronize()
buf = (C.c_ulonglong * (256 * 8 * 4))()
assert nat.lib().tron_conv_wave_stamps(buf) == 0
s = np.frombuffer(buf, dtype=np.uint64).reshape(256, 8, 4).astype(np.float64)
g = s[:, :, 3]
per = s[:, :, :3] / np.maximum(g[:, :, None], 1)
ideal = cin // 16 * 5 * 9 * (cout // 32) * 3 * 16
print(f"{cin}->{cout} B={B}: groups per workgroup {np.median(g):.0f}; per group and wave (median over workgroups):")
for w in range(8):
    print(f"  wave {w}: loop {np.median(per[:, w, 0]):7.0f} cyc, of it at barriers {np.median(per[:, w, 1]):6.0f}; epilogue + group turn {np.median(per[:, w, 2]):6.0f}")
print(f"  ideal MFMA cycles per SIMD per group {ideal}")
