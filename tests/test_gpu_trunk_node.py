"""The learner's backward through conv1..conv6 as one autograd node (Net/activations.py::_TrunkHIP, DQNNet.py:33-50):
tron_conv3x3_dgrad_mish — input gradient + residual gradient + the activation backward of the layer below + its bias
sums in one launch — against float64 autograd of the same expression, and the node against the layer-by-layer path and
against a float64 copy of the whole network."""
import copy

import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")
F = torch.nn.functional


@pytest.fixture(scope="module")
def fused():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import config  # noqa: F401
    from Net import fused
    return fused


@pytest.fixture(autouse=True)
def _layer_kernel_node(fused, monkeypatch):
    """This file holds `_TrunkHIP` — the node on the layer kernels, what f32 planes (and TRON_TRUNK_PX=0) take; int8 codes go
    to `_TrunkPX` by default (tests/test_gpu_trunk_px.py)."""
    monkeypatch.setattr(fused, "use_trunk_px", False)


def _codes(B, S, seed):
    gen = torch.Generator(device="cuda").manual_seed(seed)
    vals = torch.tensor([1, -1, -2, -3, 10, -10], dtype=torch.int8, device="cuda")
    return vals[torch.randint(0, 6, (B, S, S), device="cuda", generator=gen)]


@pytest.mark.parametrize("magnitude", [1.0, 1e-6])
@pytest.mark.parametrize("with_extra", [False, True])
@pytest.mark.parametrize("S,B,cin,cout", [(12, 1, 32, 32), (12, 37, 32, 64), (12, 131, 64, 64), (12, 1555, 32, 32),
                                          (26, 1, 32, 32), (26, 5, 32, 64), (26, 130, 64, 64)])
def test_dgrad_mish_matches_float64(fused, S, B, cin, cout, with_extra, magnitude):
    """(dgrad(gp, W) + extra) * mish'(z), its per-channel sums and maxima: relative to the largest entry, at N(0,1)-sized and
    at 1e-6-sized gradients (the scale comes from absmax, as in tron_conv3x3_dgrad); 1555: ragged last image group of a
    persistent workgroup."""
    torch.manual_seed(S + B + cin + cout)
    w = torch.randn(cout, cin, 3, 3, device="cuda") * 0.1
    gp = torch.randn(B, cout, S, S, device="cuda") * magnitude
    z = torch.randn(B, cin, S, S, device="cuda") * 2.0
    z[0, 0, 0, :4] = torch.tensor([30.0, -30.0, 21.0, -90.0], device="cuda")       # both tails of the activation
    extra = torch.randn(B, cin, S, S, device="cuda") * magnitude if with_extra else None
    zd = z.double().requires_grad_(True)
    xd = torch.zeros(B, cin, S, S, dtype=torch.float64, device="cuda", requires_grad=True)
    F.conv2d(xd, w.double(), padding=1).backward(gp.double())
    g_act = xd.grad + (extra.double() if with_extra else 0.0)
    F.mish(zd).backward(g_act)
    want = zd.grad
    absmax = gp.abs().reshape(B, -1).amax(1).contiguous()
    got, gb, am = fused.conv3x3_dgrad_mish(gp, w, absmax, z, extra)
    scale = want.abs().max().item()
    assert (got.double() - want).abs().max().item() / scale < 3e-6
    want_gb = want.sum((0, 2, 3))
    assert (gb.double() - want_gb).abs().max().item() / max(want_gb.abs().max().item(), scale) < 1e-5
    assert torch.allclose(am.double(), want.abs().amax((0, 2, 3)), rtol=1e-4, atol=0)
    # deterministic: fixed-order sums
    got2, gb2, am2 = fused.conv3x3_dgrad_mish(gp, w, absmax, z, extra)
    assert torch.equal(got, got2) and torch.equal(gb, gb2) and torch.equal(am, am2)


def test_dgrad_mish_bad_arguments(fused):
    from tron import _native as nat
    L = nat.lib()
    assert L.tron_conv3x3_dgrad_mish_workspace(8, 32, 16, 12) == 0
    assert L.tron_conv3x3_dgrad_mish_workspace(8, 32, 32, 10) == 0
    B, S = 4, 12
    gp = torch.zeros(B, 32, S, S, device="cuda")
    w = torch.zeros(32, 32, 3, 3, device="cuda")
    z = torch.zeros(B, 32, S, S, device="cuda")
    out, gb, am = torch.empty_like(z), torch.empty(32, device="cuda"), torch.empty(32, device="cuda")
    ws = torch.empty(int(L.tron_conv3x3_dgrad_mish_workspace(B, 32, 32, S)), dtype=torch.uint8, device="cuda")
    args = lambda **k: [k.get("gp", gp.data_ptr()), w.data_ptr(), None, 0, None, k.get("z", z.data_ptr()), out.data_ptr(), gb.data_ptr(),
                        am.data_ptr(), k.get("B", B), k.get("cin", 32), 32, k.get("side", S), ws.data_ptr(), None]
    assert L.tron_conv3x3_dgrad_mish(*args(gp=None)) == nat.ERR_BAD_ARG
    assert L.tron_conv3x3_dgrad_mish(*args(z=None)) == nat.ERR_BAD_ARG
    assert L.tron_conv3x3_dgrad_mish(*args(z=z.data_ptr() + 4)) == nat.ERR_BAD_ARG
    assert L.tron_conv3x3_dgrad_mish(*args(cin=48)) == nat.ERR_UNSUPPORTED
    assert L.tron_conv3x3_dgrad_mish(*args(side=14)) == nat.ERR_UNSUPPORTED
    gb.fill_(7.0)
    assert L.tron_conv3x3_dgrad_mish(*args(B=0)) == 0                  # nothing to sum: zero bias gradient
    torch.cuda.synchronize()
    assert torch.count_nonzero(gb) == 0 and torch.count_nonzero(am) == 0
    assert L.tron_conv3x3_dgrad_mish(*args()) == 0                     # a zero gradient stays zero (no scale from absmax = 0)
    torch.cuda.synchronize()
    assert torch.count_nonzero(out) == 0


@pytest.mark.parametrize("W,cin,B,codes", [(10, 3, 50, True), (10, 4, 129, False), (10, 3, 1027, True), (24, 3, 9, True), (24, 4, 5, False)])
def test_trunk_node_equals_layer_nodes(fused, W, cin, B, codes):
    """Same forward kernels, so the same output bits; parameter gradients agree with the layer-by-layer autograd graph to
    rounding (the fused epilogue evaluates mish' with exp2 / rcp, the separate pass with expf and a division)."""
    from Net.DQNNet import Net
    from tron.vec import pop_up_planes
    torch.manual_seed(W + cin + B)
    S = W + 2
    net = Net(cin, W).cuda()
    ref = copy.deepcopy(net)
    ref.fuse_trunk = False
    net.eval(), ref.eval()                                              # (dropout off: one forward each, comparable)
    c = _codes(B, S, B)
    target = torch.randn(B, 4, device="cuda")
    outs = []
    for m in (net, ref):
        if codes:
            q = m.forward_codes(c, plane4=0.25)
        else:
            x = pop_up_planes(c)
            if cin == 4:
                x = torch.cat([x, torch.full_like(x[:, :1], 0.25)], 1)
            q = m(x)
        outs.append(q)
        F.mse_loss(q, target).backward()
    if W == 10:
        assert torch.equal(outs[0], outs[1])
    else:                                                               # (conv7 at 13x13 is MIOpen's: its solver choice may differ call to call)
        assert (outs[0] - outs[1]).abs().max().item() < 1e-6
    for (name, p), (_, r) in zip(net.named_parameters(), ref.named_parameters()):
        assert p.grad is not None, name
        scale = r.grad.abs().max().item() + 1e-30
        assert (p.grad - r.grad).abs().max().item() / scale < 2e-5, name


@pytest.mark.parametrize("W,B", [(10, 40), (24, 6)])
def test_trunk_node_gradients_match_float64_network(fused, W, B):
    """Every parameter gradient of loss(Net(x)) against the same network evaluated in float64 by the plain modules."""
    from Net.DQNNet import Net
    from tron.vec import pop_up_planes
    torch.manual_seed(W)
    S = W + 2
    net = Net(3, W).cuda().eval()
    ref = copy.deepcopy(net).double()
    ref.fuse_trunk = False
    c = _codes(B, S, 3)
    target = torch.randn(B, 4, device="cuda")
    F.mse_loss(net.forward_codes(c), target).backward()
    F.mse_loss(ref._forward_plain(pop_up_planes(c).double()), target.double()).backward()
    for (name, p), (_, r) in zip(net.named_parameters(), ref.named_parameters()):
        scale = r.grad.abs().max().item() + 1e-30
        assert (p.grad.double() - r.grad).abs().max().item() / scale < 1e-4, name


def test_trunk_node_planes_gradient_and_frozen_layers(fused):
    """Planes that ask for their own gradient get it (conv1's input gradient from the library); a frozen layer gets none."""
    from Net.DQNNet import Net
    from tron.vec import pop_up_planes
    torch.manual_seed(5)
    net = Net(3, 10).cuda().eval()
    ref = copy.deepcopy(net)
    ref.fuse_trunk = False
    net.conv3.weight.requires_grad_(False)
    ref.conv3.weight.requires_grad_(False)
    x = pop_up_planes(_codes(17, 12, 1))
    xs = [x.clone().requires_grad_(True) for _ in range(2)]
    for m, xi in zip((net, ref), xs):
        m(xi).square().sum().backward()
    assert net.conv3.weight.grad is None and net.conv3.bias.grad is not None
    scale = xs[1].grad.abs().max().item()
    assert (xs[0].grad - xs[1].grad).abs().max().item() / scale < 2e-5
