"""CPU parity of the ACKTR path's host side against fixtures recorded from the reference
(tests/golden/acktr.npz; harness patches listed in make_golden.gen_acktr): the five
actor-critic nets, RolloutStorage returns, one A2C (RMSprop) update and two ACKTR (K-FAC)
updates for MapNet and Mulnet.  Net outputs within 1e-5; K-FAC factors and updated weights
within 1e-4 relative (an eigendecomposition sits in between)."""
import collections
import json
import sys
import warnings

import numpy as np
import pytest
import torch

from conftest import GOLDEN, load_golden

sys.path.insert(0, GOLDEN)
from netgen import det_state_dict  # noqa: E402

warnings.filterwarnings("ignore", message="Full backward hook is firing")


@pytest.fixture(scope="module")
def g():
    return load_golden("acktr")


def _t(a):
    return torch.from_numpy(np.asarray(a))


@pytest.mark.parametrize("name,inputs", [("MapNet", ("x4",)), ("TestNet", ("x3", "env1")), ("Net3", ("x3", "env1")),
                                         ("Net4", ("x3", "env1")), ("Mulnet", ("x3", "env2"))])
def test_actor_critic_nets_match_reference(g, name, inputs):
    import Net.ACNet as A
    shapes = collections.OrderedDict((k, tuple(v)) for k, v in json.loads(str(g["shapes_json"]))[name].items())
    net = getattr(A, name)()
    assert list(net.state_dict().keys()) == list(shapes.keys())
    assert all(tuple(v.shape) == shapes[k] for k, v in net.state_dict().items())
    net.load_state_dict(det_state_dict(shapes, salt=3))
    net.eval()
    args = [_t(g[k]) for k in inputs]
    with torch.no_grad():
        value, logits = net(*args)
        v2, logp, ent = net.evaluate_actions(args[0], _t(g[name + "_acts"]), *args[1:])
    assert np.allclose(value.numpy(), g[name + "_value"], rtol=1e-5, atol=1e-5)
    assert np.allclose(logits.numpy(), g[name + "_logits"], rtol=1e-5, atol=1e-5)
    assert np.allclose(logp.numpy(), g[name + "_logp"], rtol=1e-5, atol=1e-5)
    assert abs(float(ent) - float(g[name + "_entropy"])) < 1e-5
    a = net.act(*args)
    assert a.shape == (args[0].shape[0], 1) and int(a.min()) >= 0 and int(a.max()) <= 3
    assert torch.equal(net.deterministic_act(*args), logits.argmax(1))


def _rollouts(g, tag):
    import ACKTR
    T, N = g["r_rewards"].shape[:2]
    obs3 = g["r_obs3"]
    if tag == "map":
        obs = np.concatenate([obs3, np.full(obs3.shape[:2] + (1, 12, 12), 5.0, np.float32)], 2)
        ro = ACKTR.RolloutStorage(T, N, 4, 10, 0)
    else:
        obs = obs3
        ro = ACKTR.RolloutStorage(T, N, 3, 10, 2)
        ro.probs.copy_(_t(g["r_probs"]))
    ro.observations.copy_(_t(obs))
    ro.actions.copy_(_t(g["r_actions"]))
    ro.rewards.copy_(_t(g["r_rewards"]))
    ro.masks.copy_(_t(g["r_masks"]))
    ro.compute_returns(_t(g["r_next"]))
    return ro


def test_rollout_returns_match_reference(g):
    ro = _rollouts(g, "map")
    assert np.allclose(ro.returns.numpy(), g["returns"], rtol=1e-6, atol=1e-6)
    # insert / after_update bookkeeping (ACKTR.py:43-58)
    ro.index = 0
    for k in range(5):
        ro.insert(torch.full((16, 4, 12, 12), float(k)), torch.full((16, 1), k % 4), torch.full((16, 1), float(k)),
                  torch.ones(16, 1))
    assert ro.index == 0 and float(ro.observations[5, 0, 0, 0, 0]) == 4.0 and float(ro.rewards[2, 3]) == 2.0
    ro.after_update()
    assert torch.equal(ro.observations[0], ro.observations[-1])


PROBE = ["conv1.module.weight", "conv1.add_bias._bias", "conv7.module.weight", "fc1.module.weight",
         "actor2.module.weight", "critic3.add_bias._bias"]


@pytest.mark.parametrize("tag", ["map", "mul"])
@pytest.mark.parametrize("mode", ["a2c", "acktr"])
def test_brain_update_matches_reference(g, tag, mode):
    import ACKTR
    import Net.ACNet as A
    net = A.MapNet() if tag == "map" else A.Mulnet()
    brain = ACKTR.Brain(net, None, acktr=(mode == "acktr"), device="cpu")
    shapes = collections.OrderedDict((k, tuple(v.shape)) for k, v in net.state_dict().items())
    if mode == "acktr":                                   # checkpoint layout after the bias split
        ref_shapes = json.loads(str(g[f"{tag}_shapes_split"]))
        assert list(shapes.keys()) == list(ref_shapes.keys())
        assert all(list(shapes[k]) == ref_shapes[k] for k in shapes)
    net.load_state_dict(det_state_dict(shapes, salt=4))
    net.dropout.p = 0.0
    ro = _rollouts(g, tag)
    for k in range(2 if mode == "acktr" else 1):
        torch.manual_seed(1000 + k)
        stats = np.array([float(t) for t in brain.update(ro)])
        assert np.allclose(stats, g[f"{tag}_{mode}_stats{k}"], rtol=2e-5, atol=2e-5), (k, stats)
        sd = net.state_dict()
        for name in PROBE:
            key = name if name in sd else name.replace(".module.weight", ".weight").replace(".add_bias._bias", ".bias")
            got = sd[key].detach().numpy().reshape(-1)[:384]
            ref = g[f"{tag}_{mode}_u{k}_{name}"]
            assert np.allclose(got, ref, rtol=1e-4, atol=1e-5), (k, name, np.abs(got - ref).max())
    if mode == "acktr":
        mods = dict(net.named_modules())
        for mn in ("conv1.module", "conv7.module", "fc1.module", "actor2.add_bias"):
            for store, key in ((brain.optimizer.m_aa, "maa"), (brain.optimizer.m_gg, "mgg")):
                got = store[mods[mn]].numpy().reshape(-1)[:256]
                ref = g[f"{tag}_{key}_{mn}"]
                assert np.allclose(got, ref, rtol=1e-4, atol=1e-7 + 1e-4 * np.abs(ref).max()), (mn, key)
        assert brain.optimizer.steps == 2


@pytest.mark.parametrize("mode", ["a2c", "acktr"])
def test_micro_batched_update_equals_single_batch(g, mode):
    """Brain.update(rollouts, micro_batch=...) is the same update as the single big batch: same
    losses and same gradient; for K-FAC also the same weights after two steps.  (RMSprop's first
    steps divide by ~|g|, so A2C weights are compared through their gradients.)"""
    import ACKTR
    import Net.ACNet as A
    outs = []
    for mb in (None, 16, 33):
        net = A.Mulnet()
        brain = ACKTR.Brain(net, None, acktr=(mode == "acktr"), device="cpu")
        shapes = collections.OrderedDict((k, tuple(v.shape)) for k, v in net.state_dict().items())
        net.load_state_dict(det_state_dict(shapes, salt=4))
        net.dropout.p = 0.0
        ro = _rollouts(g, "mul")
        stats, grads = [], None
        for k in range(2 if mode == "acktr" else 1):
            torch.manual_seed(1000 + k)
            stats.append([float(t) for t in brain.update(ro, micro_batch=mb)])
            if k == 0:
                grads = torch.cat([p.grad.detach().reshape(-1) for p in net.parameters()]).numpy().copy()
        outs.append((np.array(stats), grads, torch.cat([p.detach().reshape(-1) for p in net.parameters()]).numpy()))
    for st, gr, w in outs[1:]:
        assert np.allclose(st, outs[0][0], rtol=1e-4, atol=1e-5)
        assert np.allclose(gr, outs[0][1], rtol=1e-3, atol=1e-6 + 1e-5 * np.abs(outs[0][1]).max())
        if mode == "acktr":
            assert np.allclose(w, outs[0][2], rtol=1e-4, atol=2e-6), np.abs(w - outs[0][2]).max()
