"""The learner's tail of the trunk at 24x24 / 32x32 boards — x = pool(x); x = mish(conv7(x)); x.view(-1, 64*O*O)
(Net/DQNNet.py:52-55) and its backward (DDQN.py:148) — on csrc/tron_head.hip's tron_pool_conv7_fwd / _bwd: every output
against float64 autograd of the same expression, through the C-ABI and through the network."""
import copy

import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")
F = torch.nn.functional


@pytest.fixture(scope="module")
def nat():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import config  # noqa: F401
    from tron import _native
    _native.lib()
    return _native


def _reference(x, w, b, gy):
    """float64: y, and the gradients of sum(y * gy) at x, w, b."""
    xd, wd, bd = (t.double().requires_grad_(True) for t in (x, w, b))
    pre = F.conv2d(F.avg_pool2d(xd, 3, 2, 1), wd, bd, stride=2, padding=3)
    y = F.mish(pre).reshape(x.shape[0], -1)
    y.backward(gy.double())
    return pre.detach(), y.detach(), xd.grad, wd.grad, bd.grad


def _run(nat, x, w, b, gy, want=(True, True, True)):
    L = nat.lib()
    B, side = x.shape[0], x.shape[-1]
    o = (side // 2 + 1) // 2
    dev = x.device
    saved = torch.empty(int(L.tron_pool_conv7_saved_bytes(B, side)), dtype=torch.uint8, device=dev)
    ws = torch.empty(int(L.tron_pool_conv7_workspace(B, side)), dtype=torch.uint8, device=dev)
    pre = torch.empty(B, o * o, 64, device=dev)
    y = torch.empty(B, 64 * o * o, device=dev)
    st = nat.stream_ptr()
    nat.check(L.tron_pool_conv7_fwd(nat.ptr(x), B, side, nat.ptr(w), nat.ptr(b), nat.ptr(saved), nat.ptr(pre), nat.ptr(y), nat.ptr(ws), st), "fwd")
    gx = torch.empty_like(x) if want[0] else None
    gw = torch.empty_like(w) if want[1] else None
    gb = torch.empty(64, device=dev) if want[2] else None
    nat.check(L.tron_pool_conv7_bwd(nat.ptr(gy), nat.ptr(pre), nat.ptr(saved), nat.ptr(w), B, side, nat.ptr(gx), nat.ptr(gw), nat.ptr(gb),
                                    nat.ptr(ws), st), "bwd")
    torch.cuda.synchronize()
    return pre, y, gx, gw, gb


def _rel(got, want):
    return (got.double() - want).abs().max().item() / (want.abs().max().item() + 1e-300)


@pytest.mark.parametrize("magnitude", [1.0, 1e-6])
@pytest.mark.parametrize("side,B", [(26, 1), (26, 3), (26, 37), (26, 130), (34, 1), (34, 2), (34, 41)])
def test_pool_conv7_matches_float64(nat, side, B, magnitude):
    """Forward and all three gradients, relative to the largest entry, at N(0,1)-sized and at 1e-6-sized output gradients (the
    kernels scale the gradient by a power of two taken from its largest magnitude); B = 1, 3: fewer images than workgroup slices;
    37, 41, 130: ragged slices, several images per workgroup."""
    torch.manual_seed(side * 1000 + B)
    o = (side // 2 + 1) // 2
    x = torch.randn(B, 64, side, side, device="cuda") * 1.5
    w = torch.randn(64, 64, 7, 7, device="cuda") * 0.02
    b = torch.randn(64, device="cuda") * 0.1
    gy = torch.randn(B, 64 * o * o, device="cuda") * magnitude
    x[0, 0, :2, :4] = torch.tensor([[90.0, -90.0, 40.0, -40.0], [25.0, -25.0, 60.0, -60.0]], device="cuda")
    pre_w, y_w, gx_w, gw_w, gb_w = _reference(x, w, b, gy)
    pre, y, gx, gw, gb = _run(nat, x, w, b, gy)
    assert _rel(pre.reshape(B, o, o, 64).permute(0, 3, 1, 2), pre_w) < 3e-6
    assert _rel(y, y_w) < 3e-6
    assert _rel(gx, gx_w) < 5e-6
    assert _rel(gw, gw_w) < 5e-6
    assert _rel(gb, gb_w) < 5e-6
    pre2, y2, gx2, gw2, gb2 = _run(nat, x, w, b, gy)                     # fixed-order sums: the same bits again
    assert torch.equal(y, y2) and torch.equal(gx, gx2) and torch.equal(gw, gw2) and torch.equal(gb, gb2)


def test_pool_conv7_single_taps(nat):
    """A one-hot weight tap and a one-hot output gradient: every (ky, kx) of the weight gradient and of the input gradient comes
    from the right pixels (checks the parity classes and the transposed reads' row addresses one tap at a time)."""
    torch.manual_seed(7)
    B, side, o = 2, 26, 7
    x = torch.randn(B, 64, side, side, device="cuda")
    b = torch.zeros(64, device="cuda")
    for ky, kx in [(0, 0), (0, 6), (6, 0), (6, 6), (3, 3), (2, 5), (5, 2), (1, 4)]:
        w = torch.zeros(64, 64, 7, 7, device="cuda")
        w[5, 9, ky, kx] = 1.0
        w[40, 63, 6 - ky, kx] = -0.5
        gy = torch.zeros(B, 64, o, o, device="cuda")
        gy[0, 5, ky % o, kx % o] = 1.0
        gy[1, 40, 6 - (ky % o), 3] = 2.0
        gy = gy.reshape(B, -1)
        _, y_w, gx_w, gw_w, _ = _reference(x, w, b, gy)
        _, y, gx, gw, _ = _run(nat, x, w, b, gy)
        assert _rel(y, y_w) < 3e-6, (ky, kx)
        assert _rel(gx, gx_w) < 5e-6, (ky, kx)
        assert _rel(gw, gw_w) < 5e-6, (ky, kx)


def test_pool_conv7_optional_outputs_and_bad_arguments(nat):
    L = nat.lib()
    assert L.tron_pool_conv7_workspace(8, 12) == 0 and L.tron_pool_conv7_saved_bytes(8, 24) == 0
    assert L.tron_pool_conv7_workspace(0, 26) == 0 and L.tron_pool_conv7_workspace((1 << 20) + 1, 26) == 0
    torch.manual_seed(1)
    B, side, o = 5, 26, 7
    x = torch.randn(B, 64, side, side, device="cuda")
    w = torch.randn(64, 64, 7, 7, device="cuda") * 0.02
    b = torch.randn(64, device="cuda")
    gy = torch.randn(B, 64 * o * o, device="cuda")
    full = _run(nat, x, w, b, gy)
    only_w = _run(nat, x, w, b, gy, want=(False, True, False))
    only_x = _run(nat, x, w, b, gy, want=(True, False, True))
    assert torch.equal(full[3], only_w[3]) and torch.equal(full[2], only_x[2]) and torch.equal(full[4], only_x[4])
    saved = torch.empty(int(L.tron_pool_conv7_saved_bytes(B, side)), dtype=torch.uint8, device="cuda")
    ws = torch.empty(int(L.tron_pool_conv7_workspace(B, side)), dtype=torch.uint8, device="cuda")
    pre, y = torch.empty(B, o * o, 64, device="cuda"), torch.empty(B, 64 * o * o, device="cuda")
    args = lambda **k: [k.get("x", x.data_ptr()), k.get("B", B), k.get("side", side), w.data_ptr(), b.data_ptr(), saved.data_ptr(),
                        pre.data_ptr(), y.data_ptr(), k.get("ws", ws.data_ptr()), None]
    assert L.tron_pool_conv7_fwd(*args(x=None)) == nat.ERR_BAD_ARG
    assert L.tron_pool_conv7_fwd(*args(x=x.data_ptr() + 4)) == nat.ERR_BAD_ARG
    assert L.tron_pool_conv7_fwd(*args(ws=None)) == nat.ERR_BAD_ARG
    assert L.tron_pool_conv7_fwd(*args(side=12)) == nat.ERR_UNSUPPORTED
    assert L.tron_pool_conv7_fwd(*args(B=-1)) == nat.ERR_BAD_ARG
    assert L.tron_pool_conv7_fwd(*args(B=0)) == 0
    gw, gb = torch.full((64, 64, 7, 7), 3.0, device="cuda"), torch.full((64,), 3.0, device="cuda")
    bargs = lambda **k: [k.get("gy", gy.data_ptr()), pre.data_ptr(), saved.data_ptr(), w.data_ptr(), k.get("B", B), k.get("side", side), None,
                         gw.data_ptr(), gb.data_ptr(), ws.data_ptr(), None]
    assert L.tron_pool_conv7_bwd(*bargs(gy=None)) == nat.ERR_BAD_ARG
    assert L.tron_pool_conv7_bwd(*bargs(side=30)) == nat.ERR_UNSUPPORTED
    assert L.tron_pool_conv7_bwd(*bargs(B=0)) == 0                     # nothing to sum: zero parameter gradients
    torch.cuda.synchronize()
    assert torch.count_nonzero(gw) == 0 and torch.count_nonzero(gb) == 0
    zero = torch.zeros_like(gy)                                          # a zero gradient stays zero (scale 1 from absmax = 0)
    _, _, gx, gw2, gb2 = _run(nat, x, w, b, zero)
    assert torch.count_nonzero(gx) == 0 and torch.count_nonzero(gw2) == 0 and torch.count_nonzero(gb2) == 0


def test_pool_conv7_at_the_learn_batch(nat):
    """4 096 x 26x26 (BASELINE config 3's learn step): 114 images per workgroup slice of the weight-gradient kernel, 1 568
    row tiles per parity class of the input-gradient GEMMs — against float64 on the whole batch."""
    torch.manual_seed(3)
    B, side, o = 4096, 26, 7
    x = torch.randn(B, 64, side, side, device="cuda")
    w = torch.randn(64, 64, 7, 7, device="cuda") * 0.02
    b = torch.randn(64, device="cuda") * 0.1
    gy = torch.randn(B, 64 * o * o, device="cuda") / B
    pre, y, gx, gw, gb = _run(nat, x, w, b, gy)
    gx_w = torch.empty(B, 64, side, side, dtype=torch.float64, device="cuda")
    gw_w = torch.zeros(64, 64, 7, 7, dtype=torch.float64, device="cuda")
    gb_w = torch.zeros(64, dtype=torch.float64, device="cuda")
    for i in range(0, B, 512):
        _, y_c, gx_c, gw_c, gb_c = _reference(x[i:i + 512], w, b, gy[i:i + 512])
        assert _rel(y[i:i + 512], y_c) < 3e-6
        gx_w[i:i + 512] = gx_c
        gw_w += gw_c
        gb_w += gb_c
    assert _rel(gx, gx_w) < 5e-6
    assert _rel(gw, gw_w) < 5e-6
    assert _rel(gb, gb_w) < 5e-6


@pytest.mark.parametrize("W,B", [(24, 9), (32, 3)])
def test_network_tail_on_and_off(nat, W, B, monkeypatch):
    """DQNNet.Net at 24x24 / 32x32 boards with the tail on tron_pool_conv7 and with it on the library convolution: Q agrees to
    rounding, and so does every parameter gradient; both against a float64 copy of the network."""
    from Net import activations
    from Net.DQNNet import Net
    from tron.vec import pop_up_planes
    torch.manual_seed(W)
    S = W + 2
    vals = torch.tensor([1, -1, -2, -3, 10, -10], dtype=torch.int8, device="cuda")
    codes = vals[torch.randint(0, 6, (B, S, S), device="cuda")]
    x = pop_up_planes(codes)
    target = torch.randn(B, 4, device="cuda")
    net = Net(3, W).cuda().eval()
    if W == 32:
        net.fuse_trunk = False
    ref64 = copy.deepcopy(net).double()
    ref64.fuse_trunk = False
    q64 = ref64._forward_plain(x.double())
    F.mse_loss(q64, target.double()).backward()
    results = []
    for on in (True, False):
        monkeypatch.setattr(activations, "_use_pool_conv7_cl", on)
        assert activations.pool_conv7_cl_supported(net.pool, net.conv7, torch.empty(B, 64, S, S, device="cuda")) == on
        net.zero_grad(set_to_none=True)
        q = net(x)
        F.mse_loss(q, target).backward()
        results.append((q.detach().clone(), [p.grad.clone() for p in net.parameters()]))
    assert (results[0][0].double() - q64).abs().max().item() < 1e-5
    assert (results[0][0] - results[1][0]).abs().max().item() < 1e-5
    for (name, r), g_on, g_off in zip(ref64.named_parameters(), results[0][1], results[1][1]):
        scale = r.grad.abs().max().item() + 1e-30
        assert (g_on.double() - r.grad).abs().max().item() / scale < 1e-4, name
        assert (g_on - g_off).abs().max().item() / scale < 1e-4, name


@pytest.mark.parametrize("W", [24, 32])
def test_actor_critic_tail_on_and_off(nat, W, monkeypatch):
    """The actor-critic nets share the tail (Net/ACNet.py:59-76): plain A2C (conv7 a plain module) takes tron_pool_conv7; with the
    K-FAC optimizer's SplitBias wrapper in place it keeps the hooked module.  Outputs and gradients agree with the path off."""
    from Net import activations, kfac
    from Net.ACNet import TestNet
    from tron.vec import pop_up_planes
    torch.manual_seed(W + 1)
    S, B = W + 2, 6
    vals = torch.tensor([1, -1, -2, -3, 10, -10], dtype=torch.int8, device="cuda")
    x = pop_up_planes(vals[torch.randint(0, 6, (B, S, S), device="cuda")])
    prob = torch.rand(B, device="cuda")
    net = TestNet(W).cuda().eval()
    results = []
    for on in (True, False):
        monkeypatch.setattr(activations, "_use_pool_conv7_cl", on)
        net.zero_grad(set_to_none=True)
        value, logits = net(x, prob)
        (value.square().sum() + logits.square().sum()).backward()
        results.append((value.detach().clone(), logits.detach().clone(), {n: p.grad.clone() for n, p in net.named_parameters()}))
    assert (results[0][0] - results[1][0]).abs().max().item() < 1e-5 and (results[0][1] - results[1][1]).abs().max().item() < 1e-5
    for n, g in results[0][2].items():
        scale = results[1][2][n].abs().max().item() + 1e-30
        assert (g - results[1][2][n]).abs().max().item() / scale < 1e-4, n
    monkeypatch.setattr(activations, "_use_pool_conv7_cl", True)
    kfac.split_biases(net)                                               # what KFACOptimizer does to the model (kfac.py:79-100)
    value, logits = net(x, prob)                                         # (the SplitBias branch: runs, same numbers)
    assert (value - results[1][0]).abs().max().item() < 1e-5


@pytest.mark.parametrize("with_bias", [True, False])
@pytest.mark.parametrize("ps,B", [(13, 1), (13, 37), (17, 2), (17, 41)])
def test_conv7_module_matches_float64(nat, ps, B, with_bias):
    """Net/activations.py::Conv7 — conv7 alone, NCHW in and out (tron_conv7_fwd / _bwd): the module KFACOptimizer hooks at 24x24 /
    32x32 boards, with its bias (plain A2C fallback) and without (SplitBias moved it out) — output and all gradients vs float64."""
    from Net.activations import Conv7, _Conv7HIP
    torch.manual_seed(ps * 100 + B)
    conv = Conv7(64, 64, 7, padding=3, stride=2, bias=with_bias).cuda()
    x = (torch.randn(B, 64, ps, ps, device="cuda") * 1.5).requires_grad_(True)
    y = conv(x)
    assert y.grad_fn is not None and type(y.grad_fn).__name__ == _Conv7HIP.__name__ + "Backward"
    o = (ps + 1) // 2
    assert y.shape == (B, 64, o, o)
    g = torch.randn_like(y) * 1e-3
    y.backward(g)
    xd = x.detach().double().requires_grad_(True)
    wd = conv.weight.detach().double().requires_grad_(True)
    bd = conv.bias.detach().double().requires_grad_(True) if with_bias else None
    yd = F.conv2d(xd, wd, bd, stride=2, padding=3)
    yd.backward(g.double())
    assert _rel(y.detach(), yd.detach()) < 3e-6
    assert _rel(x.grad, xd.grad) < 5e-6
    assert _rel(conv.weight.grad, wd.grad) < 5e-6
    if with_bias:
        assert _rel(conv.bias.grad, bd.grad) < 5e-6
    y2 = conv(x.detach())                                                # no graph: same forward
    assert torch.equal(y2, y.detach())
    assert type(conv(x.detach()[:, :, :12, :12].contiguous()).grad_fn).__name__ != "_Conv7HIPBackward"   # other sides: the library


def test_kfac_statistics_through_conv7_module(nat, monkeypatch):
    """K-FAC's factors for conv7 (kfac.py:41-76: input patches' Gram, output gradient's Gram) come from the module's hooks: with the
    module on tron_conv7 and on the library, the same factors and the same parameter gradients."""
    from Net import activations, kfac
    from Net.ACNet import TestNet
    from tron.vec import pop_up_planes
    torch.manual_seed(11)
    W, B = 24, 8
    S = W + 2
    vals = torch.tensor([1, -1, -2, -3, 10, -10], dtype=torch.int8, device="cuda")
    x = pop_up_planes(vals[torch.randint(0, 6, (B, S, S), device="cuda")])
    prob = torch.rand(B, device="cuda")
    base = TestNet(W).cuda().eval()
    out = []
    for on in (True, False):
        monkeypatch.setattr(activations, "_use_pool_conv7_cl", on)
        net = copy.deepcopy(base)
        opt = kfac.KFACOptimizer(net)
        opt.acc_stats = True
        value, logits = net(x, prob)
        (value.square().mean() + logits.square().mean()).backward()
        opt.acc_stats = False
        m = [mod for mod in opt.modules if isinstance(mod, torch.nn.Conv2d) and mod.kernel_size == (7, 7)][0]
        out.append((opt.m_aa[m].clone(), opt.m_gg[m].clone(), m.weight.grad.clone()))
    for a, b in zip(out[0], out[1]):
        assert (a - b).abs().max().item() / (b.abs().max().item() + 1e-30) < 1e-4
