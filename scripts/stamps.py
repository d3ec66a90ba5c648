#!/usr/bin/env python3
"""Phase timeline of k_direct from a -DTRON_STAMPS diagnostic build (development tool).
Build:  HIPCC_EXTRA=-DTRON_STAMPS csrc/build.sh -DTRON_STAMPS ; then run this on the GPU box."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "deep-q-learning_tron_amd")]
import numpy as np, torch
from tron.vec import VecTron

ed = int(os.environ.get("TRON_TILE_ENVS", "32"))
N, W = 65536, 24
env = VecTron(N, W, seed=0x5EED, obs_format="codes")
env.reset()
for _ in range(20):
    env.step(autoreset=True)
blocks = (N + ed - 1) // ed
buf = torch.zeros(blocks * 2 * 8, dtype=torch.int64, device="cuda")
env.rollout_random(1, buf)
torch.cuda.synchronize()
t = buf.cpu().numpy().reshape(blocks, 2, 8).astype(np.float64) * 0.01   # us (100 MHz)
t0 = t[:, :, 0].min()
names = ["start", "tile in LDS", "post-barrier1", "move done", "post-barrier2", "-", "-", "stream issued"]
for wv in (0, 1):
    print(f"wave {wv}: median / p10 / p90 / max (us since first block start)")
    for k, nm in enumerate(names):
        col = t[:, wv, k]
        col = col[col > 0] - t0
        if len(col):
            print(f"  {nm:14s} {np.median(col):7.2f} {np.percentile(col,10):7.2f} {np.percentile(col,90):7.2f} {col.max():7.2f}")
d = t[:, 0, 7] - t[:, 0, 0]
print("block lifetime (wave0 start->stream issued): median %.2f max %.2f" % (np.median(d), d.max()))
print("last stamp of any block: %.2f us" % (t[:, :, 7].max() - t0))
st = np.sort(t[:, 0, 0] - t0); en = np.sort(t[:, 0, 7] - t0)
print("block starts  (us) pct 0/25/50/75/90/100:", np.percentile(st, [0, 25, 50, 75, 90, 100]).round(2))
print("block ends    (us) pct 0/25/50/75/90/100:", np.percentile(en, [0, 25, 50, 75, 90, 100]).round(2))
life = t[:, 0, 7] - t[:, 0, 0]
early = t[:, 0, 0] - t0 < 1.0
print("lifetime of blocks started <1us: median %.2f ; started later: median %.2f (n=%d)" % (np.median(life[early]), np.median(life[~early]) if (~early).any() else 0, int((~early).sum())))
