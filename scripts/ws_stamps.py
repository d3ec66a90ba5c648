#!/usr/bin/env python3
"""Per-wave cycle stamps of k_conv_ws (a -DTRON_WS_STAMPS diagnostic build: scripts/ws_ablate.sh builds
scratch_bin/libtron_ws_stamps.so): where a wave's time goes per item.  usage: TRON_HIP_LIB=... ws_stamps.py [B] [S] [layer]"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "deep-q-learning_tron_amd")]
import config, torch  # noqa
import numpy as np
from Net import fused
from Net.DQNNet import Net
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
S = int(sys.argv[2]) if len(sys.argv) > 2 else 12
layer = sys.argv[3] if len(sys.argv) > 3 else "conv5"
net = Net(3, S - 2).cuda()
vals = torch.tensor([1, -1, -2, -3, 10, -10], dtype=torch.int8, device="cuda")
codes = vals[torch.randint(0, 6, (B, S, S), device="cuda")]
w = fused.ws_split_weights([net.conv2, net.conv3, net.conv4, net.conv5, net.conv6])
a = fused.conv1_px16(codes, net.conv1)
b = fused.conv_ws(a, net.conv2, w[0])
c = fused.conv_ws(b, net.conv3, w[1], residual=a)
d = fused.conv_ws(c, net.conv4, w[2])
e = fused.conv_ws(d, net.conv5, w[3])
cases = {"conv2": (a, net.conv2, w[0], None), "conv3": (b, net.conv3, w[1], a), "conv4": (c, net.conv4, w[2], None),
         "conv5": (d, net.conv5, w[3], None), "conv6": (e, net.conv6, w[4], d)}
pool = layer == "conv6pool"                            # the WS_POOL kernel (scripts/ws_ablate.sh poolstamps): "dma issue" is then the pooling pass
x, conv, wf, res = cases["conv6" if pool else layer]
for _ in range(5):
    fused.conv_ws_pool12(x, conv, wf, res) if pool else fused.conv_ws(x, conv, wf, residual=res)
torch.cuda.synchronize()
L = fused.nat.lib()
read = L.tron_conv_ws_pool_stamps if pool else L.tron_conv_ws_stamps
read.argtypes = [ctypes.c_void_p]
buf = np.zeros(256 * 12 * 8, np.uint64)
assert read(buf.ctypes.data) == 0
st = buf.reshape(256, 12, 8).astype(np.float64)       # [workgroup][wave (8 or 12 used)][stamp]
st = st[:, st[0, :, 4] > 0, :]
items = st[:, :, 4]
print(f"{layer} B={B} S={S}: items per workgroup {items.mean():.1f}")
names = ["vmcnt wait", "barrier", "dma issue", "steps"]
for i, n in enumerate(names):
    print(f"  {n:12s} per item: median {np.median(st[:, :, i] / items):9.0f} cycles   (wave 0: {np.median(st[:, 0, i] / items[:, 0]):9.0f}, wave 3: {np.median(st[:, 3, i] / items[:, 3]):9.0f})")
if pool:
    print(f"  pass + its two barriers per item: median {np.median(st[:, :, 7] / items):9.0f} cycles (inside `steps`; `dma issue` above = the pass alone)")
tot, real = st[:, :, 5], st[:, :, 6]
print(f"  total per item: {np.median(tot / items):.0f} cycles; clock {np.median(tot / real) * 100:.0f} MHz; kernel life {np.median(real) / 100:.1f} us")
