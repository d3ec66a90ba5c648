import os, sys
ROOT = "/root/repo"
sys.path[:0] = [ROOT, os.path.join(ROOT, "deep-q-learning_tron_amd"), os.path.join(ROOT, "tests")]
import config, torch
from Net import fused
F = torch.nn.functional
import test_gpu_conv_ws as T
B, S, cin, cout = int(sys.argv[1]), 12, 32, 32
torch.manual_seed(1)
conv = torch.nn.Conv2d(cin, cout, 3, padding=1).cuda()
x = torch.randn(B, cin, S, S, device="cuda")
xp = T._to_px16(fused, x)
w = fused.ws_split_weights([conv])[0]
out = fused.conv_ws(xp, conv, w, act=False, want_px=False, want_f32=True)
ref = F.conv2d(x.double(), conv.weight.double(), conv.bias.double(), padding=1)
err = (out.double() - ref).abs().amax((1, 2, 3))
bad = (err > 1e-4).nonzero().ravel()
print("bad images:", bad.numel(), bad[:40].tolist(), bad[-10:].tolist())
out2 = fused.conv_ws(xp, conv, w, act=False).float()
err = (out2.double() - ref).abs().amax((1, 2, 3))
bad = (err > 1e-4).nonzero().ravel()
print("px path bad images:", bad.numel(), bad[:40].tolist(), bad[-10:].tolist())
if bad.numel():
    i = int(bad[0]); e = (out2[i].double() - ref[i]).abs()
    print("img", i, "bad channels", (e.amax((1, 2)) > 1e-4).nonzero().ravel().tolist(), "bad rows", (e.amax((0, 2)) > 1e-4).nonzero().ravel().tolist(), "bad cols", (e.amax((0, 1)) > 1e-4).nonzero().ravel().tolist())
