#!/usr/bin/env python3
"""Net.infer wall time (HIP events) at the policy / learner batches; usage: infer_time.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "deep-q-learning_tron_amd")]
import config, torch  # noqa
from Net.DQNNet import Net
for W, B in ((10, 4096), (10, 8192), (24, 2048), (24, 4096)):
    net = Net(3, W).cuda().eval()
    S = W + 2
    codes = torch.tensor([1, -1, -2, -3, 10, -10], dtype=torch.int8, device="cuda")[torch.randint(0, 6, (B, S, S), device="cuda")]
    with torch.no_grad():
        for _ in range(5):
            net.infer(codes, codes=True)
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        torch.cuda.synchronize()
        ev[0].record()
        for _ in range(30):
            net.infer(codes, codes=True)
        ev[1].record()
        torch.cuda.synchronize()
    print(f"Net.infer {B} x {S}x{S}: {ev[0].elapsed_time(ev[1]) / 30 * 1e3:8.1f} us")
