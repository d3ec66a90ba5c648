#!/usr/bin/env python3
"""Forward+backward time of the DQN CNN at a learn batch: memory format / activation variants."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "deep-q-learning_tron_amd"))
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402
from Net.DQNNet import Net  # noqa: E402

B, W = int(sys.argv[1]) if len(sys.argv) > 1 else 4096, int(sys.argv[2]) if len(sys.argv) > 2 else 10
for name, cl, fused in (("nchw, composed mish", False, False), ("nchw, F.mish", False, True),
                        ("channels_last, F.mish", True, True)):
    torch.manual_seed(0)
    net = Net(3, W).cuda()
    net.activation = F.mish if fused else Net.mish
    x = torch.randn(B, 3, W + 2, W + 2, device="cuda")
    if cl:
        net = net.to(memory_format=torch.channels_last)
        x = x.contiguous(memory_format=torch.channels_last)
    opt = torch.optim.Adam(net.parameters())

    def step():
        opt.zero_grad(set_to_none=True)
        net(x).square().mean().backward()
        opt.step()
    for _ in range(5):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        step()
    torch.cuda.synchronize()
    print(f"{name:28s} {(time.perf_counter() - t0) / 20 * 1e3:7.2f} ms per fwd+bwd+adam at batch {B}", flush=True)
