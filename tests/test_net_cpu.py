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


def test_tensorboard_event_file(tmp_path):
    """tron/tbevents.py writes TensorBoard's on-disk format by hand (tensorboard is absent): CRC-32C
    known answers, TFRecord framing, and the Event wire encoding decoded by google.protobuf against
    message types declared here from the published event.proto / summary.proto field numbers."""
    from tron import tbevents as tb
    from tron.scalars import ScalarWriter
    assert tb.crc32c(b"123456789") == 0xE3069283            # the CRC-32C check value
    assert tb.crc32c(bytes(32)) == 0x8A9136AA               # RFC 3720 B.4: 32 bytes of zeros
    assert tb.crc32c(bytes([0xFF] * 32)) == 0x62A8AB43      # RFC 3720 B.4: 32 bytes of ones
    w = ScalarWriter(str(tmp_path / "run"))
    w.add_scalar("Training loss", 1.5, 20)
    w.add_scalar("Epsilon", 0.875, 1 << 33)
    w.close()
    ev = tb.read_events(w.events_path)                      # verifies every checksum
    assert ev[0]["file_version"] == "brain.Event:2"
    assert (ev[1]["tag"], ev[1]["value"], ev[1]["step"]) == ("Training loss", 1.5, 20)
    assert (ev[2]["tag"], ev[2]["value"], ev[2]["step"]) == ("Epsilon", 0.875, 1 << 33)
    data = open(w.events_path, "rb").read()
    with open(w.events_path, "wb") as f:                    # a flipped payload bit is caught
        f.write(data[:-6] + bytes([data[-6] ^ 1]) + data[-5:])
    with pytest.raises(ValueError):
        tb.read_events(w.events_path)

    pb = pytest.importorskip("google.protobuf")
    from google.protobuf import descriptor_pb2, descriptor_pool, message_factory
    fd = descriptor_pb2.FileDescriptorProto(name="tb_subset.proto", package="tbsub", syntax="proto3")
    T = descriptor_pb2.FieldDescriptorProto
    val = fd.message_type.add(name="Value")
    val.field.add(name="tag", number=1, type=T.TYPE_STRING, label=T.LABEL_OPTIONAL)
    val.field.add(name="simple_value", number=2, type=T.TYPE_FLOAT, label=T.LABEL_OPTIONAL)
    summ = fd.message_type.add(name="Summary")
    summ.field.add(name="value", number=1, type=T.TYPE_MESSAGE, type_name=".tbsub.Value", label=T.LABEL_REPEATED)
    evm = fd.message_type.add(name="Event")
    evm.field.add(name="wall_time", number=1, type=T.TYPE_DOUBLE, label=T.LABEL_OPTIONAL)
    evm.field.add(name="step", number=2, type=T.TYPE_INT64, label=T.LABEL_OPTIONAL)
    evm.field.add(name="file_version", number=3, type=T.TYPE_STRING, label=T.LABEL_OPTIONAL)
    evm.field.add(name="summary", number=5, type=T.TYPE_MESSAGE, type_name=".tbsub.Summary", label=T.LABEL_OPTIONAL)
    pool = descriptor_pool.DescriptorPool()
    pool.Add(fd)
    Event = message_factory.GetMessageClass(pool.FindMessageTypeByName("tbsub.Event"))
    e = Event()
    e.ParseFromString(tb.encode_scalar_event("Duration", 2.25, 123456789012, 1700000000.5))
    assert e.wall_time == 1700000000.5 and e.step == 123456789012
    assert e.summary.value[0].tag == "Duration" and e.summary.value[0].simple_value == 2.25
    e = Event()
    e.ParseFromString(tb.encode_version_event(5.0))
    assert e.file_version == "brain.Event:2" and e.wall_time == 5.0


def test_png_window(tmp_path):
    """Window(png_dir=...) writes the frame window.py:19-37 draws: reference colours and geometry."""
    import struct
    import zlib
    from tron.map import Map, Tile
    from tron.window import Window, render_rgb
    import io
    m = Map(3, 3, Tile.EMPTY, Tile.WALL)
    m[0, 0] = Tile.PLAYER_ONE_HEAD
    m[2, 1] = Tile.PLAYER_TWO_slide
    img = render_rgb(m, 10)
    assert img.shape == (50, 50, 3)
    assert tuple(img[0, 0]) == (255, 255, 255) and tuple(img[10, 10]) == (0, 0, 0)
    assert tuple(img[11, 11]) == (0, 34, 255) and tuple(img[20, 20]) == (0, 34, 255)      # head at (row 0, col 0), 0.1 offset
    assert tuple(img[31 + 5, 21 + 5]) == (250, 100, 0)                                     # P2 slide at (row 2, col 1)
    win = Window(stream=io.StringIO(), png_dir=str(tmp_path / "frames"), factor=10)
    win.render_map(m)
    data = open(tmp_path / "frames" / "frame_00001.png", "rb").read()
    assert data[:8] == b"\x89PNG\r\n\x1a\n"
    w, h, depth, ctype = struct.unpack(">IIBB", data[16:26])
    assert (w, h, depth, ctype) == (50, 50, 8, 2)
    idat_len = struct.unpack(">I", data[33:37])[0]
    assert data[37:41] == b"IDAT"
    raw = zlib.decompress(data[41:41 + idat_len])
    rows = np.frombuffer(raw, np.uint8).reshape(50, 1 + 150)
    assert np.all(rows[:, 0] == 0) and np.array_equal(rows[:, 1:].reshape(50, 50, 3), img)


@pytest.mark.parametrize("arch", ["dqn", "map", "test", "mul"])
def test_play_loads_plain_and_kfac_checkpoints(tmp_path, arch):
    """play.py:53-61 loads `.bak` files saved from Brain(net, args, acktr=True) — K-FAC's bias split renames every
    key (`conv1.module.weight`, `conv1.add_bias._bias`) — while A2C / DQN runs save the plain layout.  load_player
    recognises both and gives the same network."""
    import play
    import Net.ACNet as A
    from Net.DQNNet import Net as DQNNet
    from Net.kfac import split_biases
    torch.manual_seed(3)
    src = DQNNet(3, 10) if arch == "dqn" else {"map": A.MapNet, "test": A.TestNet, "mul": A.Mulnet}[arch]()
    plain = str(tmp_path / "plain.bak")
    torch.save(src.state_dict(), plain)
    x = torch.rand(5, 4 if arch == "map" else 3, 12, 12)
    env = {"dqn": None, "map": None, "test": torch.rand(5), "mul": torch.rand(5, 2)}[arch]

    def run(net):
        net.eval()
        with torch.no_grad():
            out = net(x) if env is None else net(x, env)
        return out if torch.is_tensor(out) else torch.cat([o.reshape(5, -1) for o in out], 1)
    want = run(src)
    a = play.load_player(plain, arch, device="cpu")
    assert torch.equal(run(a), want)
    if arch == "dqn":                                       # the DQN net is only ever trained with Adam (DDQN.py:52)
        return
    split_biases(src)                                       # what KFACOptimizer.__init__ does to the net (kfac.py:145)
    kfac = str(tmp_path / "kfac.bak")
    torch.save(src.state_dict(), kfac)
    assert any(k.endswith(".add_bias._bias") for k in src.state_dict())
    b = play.load_player(kfac, arch, device="cpu")
    assert torch.allclose(run(b), want, rtol=0, atol=1e-6)
    assert not b.training


def test_infer_rejects_misshapen_inputs():
    """The HIP kernels are handed raw pointers: Net.infer checks squareness / channel count / dtype first."""
    import pytest
    import torch
    from Net.DQNNet import Net
    net = Net(3, 10)
    with pytest.raises(TypeError):
        net.infer(torch.zeros(2, 3, 10, 12))
    with pytest.raises(TypeError):
        net.infer(torch.zeros(2, 4, 12, 12))
    with pytest.raises(TypeError):
        net.infer(torch.zeros(2, 12, 12), codes=True)                     # float codes
    with pytest.raises(TypeError):
        net.infer(torch.zeros(2, 10, 12, dtype=torch.int8), codes=True)
    assert net.infer(torch.zeros(2, 3, 12, 12)).shape == (2, 4)           # CPU tensors: the module's own forward


def test_load_player_dqn_from_kfac_layout_runs(tmp_path):
    """A DQN state_dict saved in the K-FAC key layout (conv1.module.weight / conv1.add_bias._bias) loads into a net that
    runs (the bias-split wrapper has no .weight for the HIP operators)."""
    import torch
    import play
    from Net.DQNNet import Net
    net = Net(3, 10)
    sd = {}
    for k, v in net.state_dict().items():
        if k.endswith(".weight"):
            sd[k.replace(".weight", ".module.weight")] = v
        else:
            sd[k.replace(".bias", ".add_bias._bias")] = v.reshape(-1, 1)
    path = tmp_path / "dqn_kfac.bak"
    torch.save(sd, path)
    got = play.load_player(str(path), "dqn", 10, device="cpu")
    x = torch.randn(3, 3, 12, 12)
    assert torch.allclose(got(x), net.eval()(x))


def test_conv_modules_are_plain_conv2d_off_the_gpu():
    """Net/activations.py::Conv3x3 / Conv7 are nn.Conv2d subclasses whose HIP paths apply to f32 CUDA tensors of the covered sides
    only: on the CPU (and for any other shape) they are the library convolution, with the reference's state_dict keys
    (Net/ACNet.py:59-70) — what keeps `.bak` checkpoints interchangeable."""
    import torch
    from Net.ACNet import TestNet
    from Net.activations import Conv3x3, Conv7
    torch.manual_seed(0)
    net = TestNet(24)
    assert isinstance(net.conv7, Conv7) and isinstance(net.conv7, torch.nn.Conv2d) and isinstance(net.conv2, Conv3x3)
    keys = set(net.state_dict().keys())
    assert {"conv7.weight", "conv7.bias", "conv2.weight", "conv2.bias"} <= keys and not any("module" in k for k in keys)
    x = torch.randn(3, 64, 13, 13)
    want = torch.nn.functional.conv2d(x, net.conv7.weight, net.conv7.bias, stride=2, padding=3)
    assert torch.equal(net.conv7(x), want)
    plain = torch.nn.Conv2d(64, 64, 7, padding=3, stride=2)
    plain.load_state_dict(net.conv7.state_dict())                       # same parameter names and shapes
    assert torch.equal(plain(x), want)


def test_adam_soft_on_host_tensors_is_optim_adam():
    """DDQN.AdamSoft (the device path is one HIP launch: tests/test_gpu_learner_codes.py) on host tensors takes optim.Adam's own
    step and leaves the soft update to the caller (returns False): same parameters and state bit for bit, and the state dicts
    interchange — what the CPU facade and the gloo tests run."""
    import DDQN
    torch.manual_seed(2)
    ps = [torch.nn.Parameter(torch.randn(7, 5)), torch.nn.Parameter(torch.randn(11))]
    qs = [torch.nn.Parameter(p.detach().clone()) for p in ps]
    mine, ref = DDQN.AdamSoft(ps), torch.optim.Adam(qs)
    tgt = [torch.zeros_like(p) for p in ps]
    for _ in range(3):
        for p, q in zip(ps, qs):
            p.grad = torch.randn_like(p)
            q.grad = p.grad.clone()
        assert mine.step(targets=tgt, tau=0.1) is False               # host tensors: the caller applies Agent.soft_update
        ref.step()
    assert all(torch.equal(p, q) for p, q in zip(ps, qs)) and all(float(t.abs().max()) == 0.0 for t in tgt)
    other = torch.optim.Adam(qs)
    other.load_state_dict(mine.state_dict())
    assert all(torch.equal(other.state[q]["exp_avg"], mine.state[p]["exp_avg"]) for p, q in zip(ps, qs))
