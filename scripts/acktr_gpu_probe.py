#!/usr/bin/env python3
"""How far the GPU replay of tests/golden/acktr.npz's K-FAC updates is from the recorded reference, per probed tensor,
relative to the size of the update itself; and the same with the eigendecompositions done in float64 / on the host."""
import collections, os, sys, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "deep-q-learning_tron_amd"), os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")]
warnings.filterwarnings("ignore")
import numpy as np, torch
import config  # noqa
from netgen import det_state_dict
import ACKTR, Net.ACNet as A
import Net.kfac as kfac
g = np.load(os.path.join(ROOT, "tests", "golden", "acktr.npz"))
PROBE = ["conv1.module.weight", "conv1.add_bias._bias", "conv7.module.weight", "fc1.module.weight", "actor2.module.weight", "critic3.add_bias._bias"]
t = lambda a: torch.from_numpy(np.asarray(a)).cuda()
for variant in ("gpu", ):
    net = A.MapNet()
    brain = ACKTR.Brain(net, None, acktr=True, device="cuda")
    shapes = collections.OrderedDict((k, tuple(v.shape)) for k, v in net.state_dict().items())
    sd0 = det_state_dict(shapes, salt=4)
    net.load_state_dict(sd0)
    net.dropout.p = 0.0
    T, N = g["r_rewards"].shape[:2]
    obs3 = g["r_obs3"]
    obs = np.concatenate([obs3, np.full(obs3.shape[:2] + (1, 12, 12), 5.0, np.float32)], 2)
    ro = ACKTR.RolloutStorage(T, N, 4, 10, 0, device="cuda")
    ro.observations.copy_(t(obs)); ro.actions.copy_(t(g["r_actions"])); ro.rewards.copy_(t(g["r_rewards"])); ro.masks.copy_(t(g["r_masks"]))
    ro.compute_returns(t(g["r_next"]))
    prev = {k: v.clone() for k, v in sd0.items()}
    for k in range(2):
        torch.manual_seed(1000 + k)
        stats = [float(v) for v in brain.update(ro)]
        print(variant, "update", k, "stats err", np.abs(np.array(stats) - g[f"map_acktr_stats{k}"]).max())
        sd = net.state_dict()
        for name in PROBE:
            got = sd[name].detach().cpu().numpy().reshape(-1)[:384]
            ref = g[f"map_acktr_u{k}_{name}"]
            before = prev[name].cpu().numpy().reshape(-1)[:384] if k == 0 else g[f"map_acktr_u{k-1}_{name}"]
            step = np.abs(ref - before).max()
            print(f"   {name:26s} |got-ref| {np.abs(got - ref).max():.2e}   |step| {step:.2e}   ratio {np.abs(got - ref).max() / max(step, 1e-30):.3f}")
