#!/usr/bin/env python3
"""Net.infer wall time (HIP events) at the policy / learner batches, at 12x12 with conv6 + pooling as one launch and as two
(interleaved repeats, same process); usage: infer_time.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "deep-q-learning_tron_amd")]
import config, torch  # noqa
from Net import fused
from Net.DQNNet import Net


def timed(net, codes, n=30):
    with torch.no_grad():
        for _ in range(5):
            net.infer(codes, codes=True)
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        torch.cuda.synchronize()
        ev[0].record()
        for _ in range(n):
            net.infer(codes, codes=True)
        ev[1].record()
        torch.cuda.synchronize()
    return ev[0].elapsed_time(ev[1]) / n * 1e3


for W, B in ((10, 4096), (10, 8192), (24, 2048), (24, 4096)):
    net = Net(3, W).cuda().eval()
    S = W + 2
    codes = torch.tensor([1, -1, -2, -3, 10, -10], dtype=torch.int8, device="cuda")[torch.randint(0, 6, (B, S, S), device="cuda")]
    if W == 10:
        one, two = [], []
        for _ in range(4):
            fused.use_pool_fused = True
            one.append(timed(net, codes))
            fused.use_pool_fused = False
            two.append(timed(net, codes))
        fused.use_pool_fused = True
        print(f"Net.infer {B} x {S}x{S}: {min(one):8.1f} us (two launches for conv6 + pooling: {min(two):8.1f} us)   "
              f"[{' '.join(f'{a:.0f}/{b:.0f}' for a, b in zip(one, two))}]")
    else:
        print(f"Net.infer {B} x {S}x{S}: {timed(net, codes):8.1f} us")
