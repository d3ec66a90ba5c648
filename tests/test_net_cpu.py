"""CPU checks of the learner's host side against fixtures recorded from the reference
(tests/golden/net.npz: Q-values of the patched DQNNet.Net and one real DDQN Agent.learn()).
Tolerance 1e-5, as BASELINE.json's north_star states for Q-values."""
import collections
import json
import os
import sys

import numpy as np
import pytest
import torch

from conftest import GOLDEN, load_golden

sys.path.insert(0, GOLDEN)
from netgen import det_state_dict  # noqa: E402

TOL = 1e-5


@pytest.fixture(scope="module")
def fx():
    g = load_golden("net")
    shapes = collections.OrderedDict((k, tuple(v)) for k, v in json.loads(str(g["shapes_json"])).items())
    return g, shapes


def test_state_dict_layout_matches_reference(fx):
    g, shapes = fx
    from Net.DQNNet import Net
    sd = Net(4, 10).state_dict()
    assert list(sd.keys()) == list(shapes.keys())                 # same 22 names, same order (.bak interchange)
    assert all(tuple(sd[k].shape) == shapes[k] for k in shapes)
    assert sum(v.numel() for v in sd.values()) == 501924


def test_q_values_match_reference(fx):
    g, shapes = fx
    from Net.DQNNet import Net
    net = Net(4, 10)
    net.load_state_dict(det_state_dict(shapes, salt=0))
    net.eval()
    with torch.no_grad():
        q = net(torch.from_numpy(g["x"])).numpy()
    assert np.allclose(q, g["q"], rtol=TOL, atol=TOL), np.abs(q - g["q"]).max()
    assert np.array_equal(net.act(torch.from_numpy(g["x"])).numpy(), g["q"].argmax(1))


def test_ddqn_learn_step_matches_reference(fx):
    """Same weights, same batch, dropout off: loss, Adam-updated local net and soft-updated
    target net after ONE Agent.learn() equal the reference's (DDQN.py:115-165)."""
    g, shapes = fx
    import DDQN
    agent = DDQN.Agent(10, 4, device="cpu", make_memory=False)
    agent.qnetwork_local.load_state_dict(det_state_dict(shapes, salt=1))
    agent.qnetwork_target.load_state_dict(det_state_dict(shapes, salt=2))
    agent.qnetwork_local.dropout.p = 0.0
    agent.qnetwork_target.dropout.p = 0.0
    exp = tuple(torch.from_numpy(g[k]) for k in ("ls", "la", "lr", "ls2", "ld"))
    with torch.no_grad():
        qb = agent.qnetwork_local.eval()(exp[0]).numpy()
    assert np.allclose(qb, g["q_local_before"], rtol=TOL, atol=1e-4)
    assert DDQN.GAMMA == float(g["gamma"]) and DDQN.TAU == float(g["tau"])
    loss = float(agent.learn(exp, DDQN.GAMMA))
    assert abs(loss - float(g["loss"])) <= 1e-5 * abs(float(g["loss"]))
    for net, tag in ((agent.qnetwork_local, "local_"), (agent.qnetwork_target, "target_")):
        sd = net.state_dict()
        for k in ("conv1.weight", "conv4.bias", "conv7.weight", "fc1.weight", "actor2.weight", "actor2.bias"):
            got = sd[k].detach().numpy().reshape(-1)[:512]
            assert np.allclose(got, g[tag + k], rtol=TOL, atol=TOL), (tag, k, np.abs(got - g[tag + k]).max())


def test_planes_to_codes_inverts_pop_up():
    import DDQN
    e = load_golden("encode")
    codes = torch.from_numpy(e["codes_10"].reshape(-1, 12, 12))
    planes = torch.from_numpy(e["planes_10"].reshape(-1, 3, 12, 12)).float()
    assert torch.equal(DDQN.planes_to_codes(planes), codes)


def test_net_sizes_for_larger_boards():
    from Net.DQNNet import Net, conv7_side
    assert conv7_side(12) == 3 and conv7_side(26) == 7 and conv7_side(34) == 9
    n = Net(3, 24)
    assert n.fc1.in_features == 3136
    assert n(torch.zeros(5, 3, 26, 26)).shape == (5, 4)


def test_dqn_learn_step_smooth_l1():
    """DQN.py:262-292 target rule: y = r if terminal else r + gamma max Q(s')."""
    import DQN
    torch.manual_seed(1)
    model = DQN.Net(in_channels=1, width=10)
    model.dropout.p = 0.0
    opt = torch.optim.Adam(model.parameters())
    s, s2 = torch.randn(6, 1, 12, 12), torch.randn(6, 1, 12, 12)
    a = torch.randint(0, 4, (6, 1))
    r = torch.tensor([0., 1., 2., 100., -25., 0.])
    term = torch.tensor([0, 0, 0, 1, 1, 1])
    with torch.no_grad():
        exp = torch.nn.functional.smooth_l1_loss(model(s).gather(1, a).sum(1),
                                                 r + 0.9 * model(s2).max(1)[0] * (1 - term.float()))
    loss = DQN.learn_step(model, opt, s, a, s2, r, term)
    assert torch.allclose(loss, exp, rtol=1e-5, atol=1e-6)


def test_checkpoint_roundtrip_and_scalars(tmp_path):
    import DDQN
    from tron.scalars import ScalarWriter, read_scalars
    torch.manual_seed(0)
    a = DDQN.Agent(10, 3, device="cpu", make_memory=False)
    exp = (torch.rand(8, 3, 12, 12), torch.randint(0, 4, (8, 1)), torch.randn(8, 1), torch.rand(8, 3, 12, 12),
           torch.zeros(8, 1))
    a.qnetwork_local.dropout.p = 0.0
    a.learn(exp, 0.9)
    path = str(tmp_path / "ddqn.full")
    DDQN.save_checkpoint(path, a, epsilon=0.42, counters={"games": 7})
    b = DDQN.Agent(10, 3, device="cpu", make_memory=False)
    eps, counters = DDQN.load_checkpoint(path, b)
    assert eps == 0.42 and counters == {"games": 7}
    b.qnetwork_local.dropout.p = 0.0
    la, lb = a.learn(exp, 0.9), b.learn(exp, 0.9)          # same next step => optimizer state restored too
    assert torch.allclose(la, lb) and all(torch.equal(p, q) for p, q in
                                          zip(a.qnetwork_local.parameters(), b.qnetwork_local.parameters()))
    w = ScalarWriter(str(tmp_path / "run"))
    w.add_scalar("Training loss", 1.5, 20)
    w.close()
    recs = read_scalars(w.path)
    assert recs[0]["tag"] == "Training loss" and recs[0]["value"] == 1.5 and recs[0]["step"] == 20


def test_ascii_window():
    import io
    from tron.map import Map, Tile
    from tron.window import Window, render_ascii
    m = Map(3, 3, Tile.EMPTY, Tile.WALL)
    m[0, 0] = Tile.PLAYER_ONE_HEAD
    m[2, 1] = Tile.PLAYER_TWO_slide
    assert render_ascii(m).split("\n") == ["#####", "#A..#", "#...#", "#.-.#", "#####"]
    buf = io.StringIO()
    win = Window(stream=buf)
    win.render_map(m)
    assert win.frames == 1 and "#A..#" in buf.getvalue()
