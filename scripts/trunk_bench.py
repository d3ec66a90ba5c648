#!/usr/bin/env python3
"""Time the conv trunk (conv1..conv6) of the DQN net per arithmetic mode; usage: trunk_bench.py [B] [S]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "deep-q-learning_tron_amd")]
import config, torch  # noqa
from Net import fused
from Net.DQNNet import Net
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
S = int(sys.argv[2]) if len(sys.argv) > 2 else 12
net = Net(3, S - 2).cuda()
vals = torch.tensor([1, -1, -2, -3, 10, -10], dtype=torch.int8, device="cuda")
codes = vals[torch.randint(0, 6, (B, S, S), device="cuda")]
fl = 2 * B * S * S * 9 * (3 * 32 + 2 * 32 * 32 + 32 * 64 + 2 * 64 * 64)
for math in ("f32", "f16x3"):
    for _ in range(3):
        fused.trunk(net, codes, codes=True, math=math)
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(10):
        fused.trunk(net, codes, codes=True, math=math)
    ev[1].record()
    torch.cuda.synchronize()
    t = ev[0].elapsed_time(ev[1]) / 10
    print(f"trunk {math} B={B} S={S}: {t:.3f} ms = {fl / t / 1e9:.1f} TF/s f32-equivalent", flush=True)
t0 = time.perf_counter()
q = net.infer(codes, codes=True)
torch.cuda.synchronize()
for _ in range(5):
    q = net.infer(codes, codes=True)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    q = net.infer(codes, codes=True)
torch.cuda.synchronize()
print(f"Net.infer: {(time.perf_counter() - t0) / 10 * 1e3:.3f} ms")
if fused.head_supported(net, S):
    x = fused.trunk(net, codes, codes=True)

    def lib_tail(x):
        with torch.no_grad():
            y = torch.nn.functional.mish(net.conv7(net.pool(x))).reshape(-1, net.flat)
            y = torch.nn.functional.mish(net.fc1(y))
            y = torch.nn.functional.mish(net.fc2(y))
            return net.actor2(torch.nn.functional.mish(net.actor1(y)))
    for name, fn in (("tron_dqn_head_fwd", lambda: fused.head(net, x)), ("library tail", lambda: lib_tail(x))):
        for _ in range(3):
            fn()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        ev[0].record()
        for _ in range(10):
            fn()
        ev[1].record()
        torch.cuda.synchronize()
        print(f"{name}: {ev[0].elapsed_time(ev[1]) / 10:.3f} ms", flush=True)
    print("head vs library tail max |dQ|:", (fused.head(net, x) - lib_tail(x)).abs().max().item())
