"""csrc/tron_conv.hip — the CNN's 3x3 convolutions on the fp32 matrix cores — against a float64 torch reference
of the same op (tolerance 1e-5, the north star's bound for Q-values), and the whole inference path
(Net.infer: HIP convolutions + library tail) against the Q-values recorded from the reference network
(tests/golden/net.npz, DQNNet.py:33-63)."""
import collections
import json
import sys

import numpy as np
import pytest

from conftest import GOLDEN, load_golden

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")
F = torch.nn.functional
TOL = 1e-5


@pytest.fixture(scope="module")
def fused():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import config  # noqa: F401
    from Net import fused
    return fused


def _ref(x, conv, res, act):
    y = F.conv2d(x.double(), conv.weight.double(), conv.bias.double(), padding=1)
    if res is not None:
        y = y + res.double()
    return y, (F.mish(y) if act else y)


@pytest.mark.parametrize("math", ["f32", "f16x3"])
@pytest.mark.parametrize("S,B", [(12, 1), (12, 7), (12, 260), (12, 1555), (26, 1), (26, 3), (26, 130), (26, 301)])   # 1555: several image groups per persistent workgroup, ragged
@pytest.mark.parametrize("cin,cout", [(32, 32), (32, 64), (64, 64), (64, 32)])
def test_conv3x3_matches_float64_reference(fused, S, B, cin, cout, math, monkeypatch):
    """Both arithmetic modes of tron_conv3x3_fwd: the exact-f32 MFMA kernel and the split-f16 one (12x12 boards; at
    26x26 the request falls back to the f32 kernel inside the library)."""
    monkeypatch.setattr(fused, "default_math", math)
    torch.manual_seed(S * 1000 + B + cin + cout)
    conv = torch.nn.Conv2d(cin, cout, 3, padding=1).cuda()
    x = torch.randn(B, cin, S, S, device="cuda")
    res = torch.randn(B, cout, S, S, device="cuda")
    for r, act in ((res, True), (None, True), (res, False)):
        got, pre = fused.conv3x3(x, conv, residual=r, act=act, want_pre=True)
        ref_pre, ref = _ref(x, conv, r, act)
        assert (pre.double() - ref_pre).abs().max().item() < TOL
        assert (got.double() - ref).abs().max().item() < TOL
    # asymmetric weights / one-hot inputs: a transposed tap or a swapped row/column cannot hide
    with torch.no_grad():
        conv.weight.copy_(torch.arange(conv.weight.numel(), device="cuda").reshape(conv.weight.shape).float() % 17 - 8)
        conv.bias.zero_()
    x = torch.zeros(B, cin, S, S, device="cuda")
    x[:, 1, 2, 3] = 1.0
    x[:, cin - 1, S - 1, 0] = 2.0
    got = fused.conv3x3(x, conv, act=False)
    assert torch.equal(got, F.conv2d(x, conv.weight, None, padding=1))       # small integers: exact


@pytest.mark.parametrize("math", ["f32", "f16x3"])
@pytest.mark.parametrize("S,B", [(12, 5), (12, 1030), (26, 2), (26, 300)])
@pytest.mark.parametrize("cin", [3, 4])
def test_conv1_from_codes_and_from_planes(fused, S, B, cin, math, monkeypatch):
    """conv1 reads the env's int8 observation codes (map.py:67-84) and builds util.pop_up's planes
    (util.py:11-37) (+ the constant prob_map plane, game.py:124-132) on the fly."""
    from tron.vec import pop_up_planes
    monkeypatch.setattr(fused, "default_math", math)
    torch.manual_seed(S + B + cin)
    conv = torch.nn.Conv2d(cin, 32, 3, padding=1).cuda()
    vals = torch.tensor([1, -1, -2, -3, 10, -10], dtype=torch.int8, device="cuda")
    codes = vals[torch.randint(0, 6, (B, S, S), device="cuda")]
    planes = pop_up_planes(codes)
    if cin == 4:
        planes = torch.cat([planes, torch.full((B, 1, S, S), 5.0, device="cuda")], 1)
    _, ref = _ref(planes, conv, None, True)
    got_c = fused.conv3x3(codes, conv, codes=True, plane4=5.0)
    got_p = fused.conv3x3(planes, conv)
    assert (got_c.double() - ref).abs().max().item() < TOL
    assert torch.equal(got_c, got_p)                       # same arithmetic, two ways of staging the input


def test_unsupported_shapes_are_reported(fused):
    from tron import _native as nat
    conv = torch.nn.Conv2d(32, 32, 3, padding=1).cuda()
    assert fused.supported(conv, 12) and fused.supported(conv, 26) and fused.supported(conv, 34) and not fused.supported(conv, 13)
    assert not fused.supported(torch.nn.Conv2d(32, 48, 3, padding=1).cuda(), 12)
    with pytest.raises(nat.TronNativeError):
        fused.conv3x3(torch.randn(2, 32, 14, 14, device="cuda"), conv)
    # activations far outside f16's range go through the split kernel unharmed (2^-6 pre-scale, exact)
    x = torch.randn(6, 32, 12, 12, device="cuda") * 3.0e5
    got = fused.conv3x3(x, conv, act=False, math="f16x3")
    ref = F.conv2d(x.double(), conv.weight.double(), conv.bias.double(), padding=1)
    assert torch.isfinite(got).all() and ((got.double() - ref).abs().max() / ref.abs().max()).item() < 1e-6


@pytest.mark.parametrize("math", ["f32", "f16x3"])
def test_infer_matches_reference_q_values(fused, math, monkeypatch):
    """Net.infer (HIP trunk, either arithmetic) on the recorded inputs: Q within 1e-5 of the reference network's."""
    monkeypatch.setattr(fused, "default_math", math)
    sys.path.insert(0, GOLDEN)
    from netgen import det_state_dict
    from Net.DQNNet import Net
    g = load_golden("net")
    shapes = collections.OrderedDict((k, tuple(v)) for k, v in json.loads(str(g["shapes_json"])).items())
    net = Net(4, 10).cuda()
    net.load_state_dict(det_state_dict(shapes, salt=0))
    x = torch.from_numpy(g["x"]).cuda()
    q = net.infer(x).cpu().numpy()
    assert np.allclose(q, g["q"], rtol=TOL, atol=TOL), np.abs(q - g["q"]).max()
    assert np.array_equal(q.argmax(1), g["q"].argmax(1))
    assert net.training                                     # infer() leaves the module's mode alone


@pytest.mark.parametrize("W,B,cin", [(10, 513, 3), (10, 64, 4), (24, 37, 3)])
def test_infer_equals_module_forward_and_codes_path(fused, W, B, cin):
    from Net.DQNNet import Net
    from tron.vec import pop_up_planes
    torch.manual_seed(W + B)
    S = W + 2
    net = Net(cin, W).cuda()
    vals = torch.tensor([1, -1, -2, -3, 10, -10], dtype=torch.int8, device="cuda")
    codes = vals[torch.randint(0, 6, (B, S, S), device="cuda")]
    planes = pop_up_planes(codes)
    if cin == 4:
        planes = torch.cat([planes, torch.full((B, 1, S, S), 5.0, device="cuda")], 1)
    net.eval()
    with torch.no_grad():
        ref = net.double()(planes.double())
    net.float().train()
    q_p = net.infer(planes)
    q_c = net.infer(codes, codes=True, plane4=5.0)
    assert (q_p.double() - ref).abs().max().item() < TOL
    assert (q_c.double() - ref).abs().max().item() < TOL       # (codes: the weight-stationary chain; planes: the layer kernels)
    fused.use_ws = False
    try:
        assert torch.equal(q_p, net.infer(codes, codes=True, plane4=5.0))   # same kernels, two ways of staging conv1's input
    finally:
        fused.use_ws = True


@pytest.mark.parametrize("S,B,cin,cout,with_res", [(12, 33, 32, 32, True), (12, 64, 32, 64, False), (12, 20, 64, 64, True),
                                                   (26, 9, 32, 32, True), (12, 40, 3, 32, False), (12, 40, 4, 32, False)])
def test_training_conv_gradients_match_float64(fused, S, B, cin, cout, with_res):
    """The differentiable path (Net/activations.py::_ConvBiasMishHIP): forward on tron_conv3x3_fwd, input gradient on
    the same kernel reading the weight transposed and tap-flipped (tron_conv3x3_dgrad), activation + bias gradient in
    tron_bias_mish_bwd, weight gradient on tron_conv3x3_wgrad (12x12; 26x26 on MIOpen) — against autograd through the
    float64 composition mish(conv2d(x) + residual)."""
    from Net.activations import conv_bias_mish
    torch.manual_seed(S + B + cin + cout)
    conv = torch.nn.Conv2d(cin, cout, 3, padding=1).cuda()
    x = torch.randn(B, cin, S, S, device="cuda", requires_grad=cin >= 8)
    res = torch.randn(B, cout, S, S, device="cuda", requires_grad=True) if with_res else None
    gout = torch.randn(B, cout, S, S, device="cuda")
    out = conv_bias_mish(conv, x, res)
    assert type(out.grad_fn).__name__ == "_ConvBiasMishHIPBackward"
    out.backward(gout)
    xd = x.detach().double().requires_grad_(cin >= 8)
    rd = res.detach().double().requires_grad_(True) if with_res else None
    wd, bd = conv.weight.detach().double().requires_grad_(True), conv.bias.detach().double().requires_grad_(True)
    y = F.conv2d(xd, wd, bd, padding=1)
    ref = F.mish(y + rd if with_res else y)
    ref.backward(gout.double())
    assert (out.double() - ref).abs().max().item() < TOL

    def close(a, b, what):
        scale = max(1.0, b.abs().max().item())
        assert (a.double() - b).abs().max().item() < 2e-5 * scale, (what, (a.double() - b).abs().max().item(), scale)
    close(conv.weight.grad, wd.grad, "weight")
    close(conv.bias.grad, bd.grad, "bias")
    if cin >= 8:
        close(x.grad, xd.grad, "input")
    if with_res:
        close(res.grad, rd.grad, "residual")


@pytest.mark.parametrize("S,B", [(12, 1), (12, 131), (12, 1031), (26, 2), (26, 65)])
def test_split16_chain_equals_unchained_layers(fused, S, B, monkeypatch):
    """Layers chained through the split-f16 image (TRON_CONV_IN_SPLIT16 / out_split) give bit-for-bit what the same
    split kernel gives when every layer re-splits the previous layer's f32 output: the image IS that split.
    (The chunked kernels of csrc/tron_conv_f16.hip: the weight-stationary chain, which fused.trunk prefers for codes,
    is switched off here and has its own tests in test_gpu_conv_ws.py.)"""
    from Net.DQNNet import Net
    monkeypatch.setattr(fused, "use_ws", False)
    torch.manual_seed(S + B)
    net = Net(3, S - 2).cuda()
    vals = torch.tensor([1, -1, -2, -3, 10, -10], dtype=torch.int8, device="cuda")
    codes = vals[torch.randint(0, 6, (B, S, S), device="cuda")]
    chained = fused.trunk(net, codes, codes=True, math="f16x3")
    x = fused.conv3x3(codes, net.conv1, codes=True, math="f16x3")
    idx = x
    x = fused.conv3x3(x, net.conv2, math="f16x3")
    x = fused.conv3x3(x, net.conv3, residual=idx, math="f16x3")
    x = fused.conv3x3(x, net.conv4, math="f16x3")
    idx = x
    x = fused.conv3x3(x, net.conv5, math="f16x3")
    x = fused.conv3x3(x, net.conv6, residual=idx, math="f16x3")
    assert torch.equal(chained, x)
    # and a layer asked for both forms returns the same f32 tensor either way
    a = fused.conv3x3(idx, net.conv5, math="f16x3")
    b, s16 = fused.conv3x3(idx, net.conv5, math="f16x3", want_split=True)
    assert torch.equal(a, b) and s16.buf.numel() == a.numel() * 4


def test_dqn_head_26_matches_float64_reference(fused):
    """The 24x24 boards' head: pooled 13x13 planes, conv7 as the implicit GEMM over its 49 taps, fc1 on 64*7*7."""
    from Net.DQNNet import Net
    for B in (1, 3, 50, 131):
        torch.manual_seed(B)
        net = Net(3, 24).cuda()
        x = torch.randn(B, 64, 26, 26, device="cuda") * 1.5
        assert fused.head_supported(net, 26)
        q, g = fused.head(net, x, want_greedy=True)
        ref = _head_ref(net, x)
        assert (q.double() - ref).abs().max().item() < TOL, (B, (q.double() - ref).abs().max().item())
        assert torch.equal(g.long(), q.argmax(1))


def _head_ref(net, x):
    d = lambda t: t.double()
    y = F.avg_pool2d(x.double(), 3, stride=2, padding=1)
    y = F.mish(F.conv2d(y, d(net.conv7.weight), d(net.conv7.bias), stride=2, padding=3)).reshape(x.shape[0], -1)
    y = F.mish(F.linear(y, d(net.fc1.weight), d(net.fc1.bias)))
    y = F.mish(F.linear(y, d(net.fc2.weight), d(net.fc2.bias)))
    return F.linear(F.mish(F.linear(y, d(net.actor1.weight), d(net.actor1.bias))), d(net.actor2.weight), d(net.actor2.bias))


@pytest.mark.parametrize("B", [1, 5, 127, 128, 129, 1000, 4099])
def test_dqn_head_matches_float64_reference(fused, B):
    """csrc/tron_head.hip (pool + dense conv7 + fc1 + fc2 + actor1 + actor2, DQNNet.py:52-63) on a random trunk output:
    Q within 1e-5 of float64 torch, greedy action = argmax of the Q it returns; ragged batches exercise the row clamp."""
    from Net.DQNNet import Net
    torch.manual_seed(B)
    net = Net(4, 10).cuda()
    x = torch.randn(B, 64, 12, 12, device="cuda") * 1.5
    assert fused.head_supported(net, 12)
    q, g = fused.head(net, x, want_greedy=True)
    ref = _head_ref(net, x)
    assert (q.double() - ref).abs().max().item() < TOL, (q.double() - ref).abs().max().item()
    assert torch.equal(g.long(), q.argmax(1))
    only_g = fused.head(net, x, want_q=False, want_greedy=True)[1]
    assert torch.equal(only_g, g)


def test_dqn_head_large_activations_and_bad_args(fused):
    """Trained nets are not N(0,1): activations up to ~100 and weights 4x the init scale keep the relative error."""
    from Net.DQNNet import Net
    from tron import _native as nat
    torch.manual_seed(3)
    net = Net(4, 10).cuda()
    with torch.no_grad():
        for p in net.parameters():
            p.mul_(2.0)
    x = torch.randn(300, 64, 12, 12, device="cuda") * 30
    q = fused.head(net, x)
    ref = _head_ref(net, x)
    assert torch.isfinite(q).all() and ((q.double() - ref).abs().max() / ref.abs().max()).item() < 1e-6
    L = nat.lib()
    assert L.tron_dqn_head_workspace(16, 34) == 0
    ws = torch.empty(int(L.tron_dqn_head_workspace(4, 12)), dtype=torch.uint8, device="cuda")
    args = [net.conv7.weight, net.conv7.bias, net.fc1.weight, net.fc1.bias, net.fc2.weight, net.fc2.bias,
            net.actor1.weight, net.actor1.bias, net.actor2.weight, net.actor2.bias]
    ptrs = [t.data_ptr() for t in args]
    qo = torch.empty(4, 4, device="cuda")
    assert L.tron_dqn_head_fwd(x.data_ptr(), 4, 34, *ptrs, ws.data_ptr(), qo.data_ptr(), None, None) == nat.ERR_UNSUPPORTED
    assert L.tron_dqn_head_fwd(x.data_ptr(), 4, 12, *ptrs, ws.data_ptr(), None, None, None) == nat.ERR_BAD_ARG
    assert L.tron_dqn_head_fwd(x.data_ptr(), 4, 12, *ptrs, None, qo.data_ptr(), None, None) == nat.ERR_BAD_ARG
    assert L.tron_dqn_head_fwd(x.data_ptr(), 0, 12, *ptrs, ws.data_ptr(), qo.data_ptr(), None, None) == 0


@pytest.mark.parametrize("W", [10, 24])
def test_infer_greedy_is_argmax_of_infer(fused, W):
    from Net.DQNNet import Net
    torch.manual_seed(W)
    net = Net(3, W).cuda()
    vals = torch.tensor([1, -1, -2, -3, 10, -10], dtype=torch.int8, device="cuda")
    codes = vals[torch.randint(0, 6, (257, W + 2, W + 2), device="cuda")]
    g = net.infer(codes, codes=True, greedy=True)
    assert g.dtype == torch.int8 and torch.equal(g.long(), net.infer(codes, codes=True).argmax(1))


@pytest.mark.parametrize("magnitude", [1.0, 1e-6, 3e4])
@pytest.mark.parametrize("B", [1, 2, 3, 257, 700])
@pytest.mark.parametrize("cin,cout", [(3, 32), (4, 32), (32, 32), (32, 64), (64, 64), (64, 32), (4, 64)])
def test_conv3x3_wgrad_matches_float64(fused, cin, cout, B, magnitude):
    """tron_conv3x3_wgrad against the float64 weight gradient of F.conv2d, at gradient magnitudes from 1e-6 (what a
    mean-reduced loss over a 4 096 batch hands down) to 3e4: relative error below 2e-6 of the largest entry, with the
    scale taken from a pre-pass (absmax=None) and from tron_bias_mish_bwd-style block maxima."""
    torch.manual_seed(B * 131 + cin + cout)
    x = torch.randn(B, cin, 12, 12, device="cuda")
    gp = torch.randn(B, cout, 12, 12, device="cuda") * magnitude
    gp[0, 0, 0, 0] = 0.0
    wd = torch.zeros(cout, cin, 3, 3, dtype=torch.float64, device="cuda", requires_grad=True)
    F.conv2d(x.double(), wd, padding=1).backward(gp.double())
    ref = wd.grad
    blocks = gp.abs().reshape(-1)[: (gp.numel() // 7) * 7].reshape(7, -1).amax(1).contiguous()    # any partition's maxima do
    blocks = torch.maximum(blocks, gp.abs().max().expand(7) * (torch.arange(7, device="cuda") == 3))
    for absmax in (None, blocks):
        got = fused.conv3x3_wgrad(x, gp, absmax)
        err = (got.double() - ref).abs().max().item() / ref.abs().max().item()
        assert err < 2e-6, (err, absmax is None)
    assert torch.equal(fused.conv3x3_wgrad(x, gp, blocks), got)                    # deterministic


def test_conv3x3_wgrad_zero_gradient_and_bad_args(fused):
    from tron import _native as nat
    L = nat.lib()
    x = torch.randn(5, 32, 12, 12, device="cuda")
    gp = torch.zeros(5, 32, 12, 12, device="cuda")
    assert torch.count_nonzero(fused.conv3x3_wgrad(x, gp)) == 0
    gw = torch.empty(32, 32, 3, 3, device="cuda")
    ws = torch.empty(int(L.tron_conv3x3_wgrad_workspace(32, 32)), dtype=torch.uint8, device="cuda")
    a = (x.data_ptr(), gp.data_ptr(), None, 0, gw.data_ptr(), 5)
    assert L.tron_conv3x3_wgrad(*a, 32, 32, 30, ws.data_ptr(), None) == nat.ERR_UNSUPPORTED
    assert L.tron_conv3x3_wgrad(*a, 64, 32, 26, ws.data_ptr(), None) == nat.ERR_UNSUPPORTED
    assert L.tron_conv3x3_wgrad(*a, 48, 32, 12, ws.data_ptr(), None) == nat.ERR_UNSUPPORTED
    assert L.tron_conv3x3_wgrad(*a, 32, 32, 12, None, None) == nat.ERR_BAD_ARG
    assert L.tron_conv3x3_wgrad(x.data_ptr(), gp.data_ptr(), None, 0, gw.data_ptr(), 0, 32, 32, 12, ws.data_ptr(), None) == 0
    torch.cuda.synchronize()
    assert torch.count_nonzero(gw) == 0                                            # batch 0: the sum over nothing


@pytest.mark.parametrize("magnitude", [1.0, 1e-7])
@pytest.mark.parametrize("S,B,cin,cout", [(12, 37, 32, 32), (12, 5, 32, 64), (12, 130, 64, 64), (26, 3, 64, 32)])
def test_conv3x3_dgrad_matches_float64_at_gradient_magnitudes(fused, S, B, cin, cout, magnitude):
    """tron_conv3x3_dgrad scales the gradient by the power of two its block maxima give: 1e-7-sized gradients keep
    the relative accuracy that N(0, 1) ones have (the activations' fixed 2^-6 would put them in f16's denormals)."""
    torch.manual_seed(S + B + cin)
    w = torch.randn(cout, cin, 3, 3, device="cuda") * 0.1
    gp = torch.randn(B, cout, S, S, device="cuda") * magnitude
    xd = torch.zeros(B, cin, S, S, dtype=torch.float64, device="cuda", requires_grad=True)
    F.conv2d(xd, w.double(), padding=1).backward(gp.double())
    absmax = gp.abs().reshape(B, -1).amax(1).contiguous()
    got = fused.conv3x3_dgrad(gp, w, absmax)
    err = (got.double() - xd.grad).abs().max().item() / xd.grad.abs().max().item()
    assert err < 2e-6, err


@pytest.mark.parametrize("B", [1, 7, 300])
def test_pool_conv7_dense_path_matches_float64(fused, B):
    """Net/activations.py::_PoolConv7 (tron_pool12 + conv7 as GEMMs on its dense form, DQNNet.py:52-55) against autograd
    through float64 mish(conv2d(avg_pool2d(x))): output, input gradient, weight and bias gradients."""
    from Net.activations import pool_conv7_mish, pool_conv7_supported
    torch.manual_seed(B)
    pool = torch.nn.AvgPool2d(kernel_size=3, padding=1, stride=2)
    conv = torch.nn.Conv2d(64, 64, 7, padding=3, stride=2).cuda()
    x = torch.randn(B, 64, 12, 12, device="cuda", requires_grad=True)
    assert pool_conv7_supported(pool, conv, x)
    out = pool_conv7_mish(pool, conv, x)
    gout = torch.randn_like(out)
    out.backward(gout)
    xd = x.detach().double().requires_grad_(True)
    wd, bd = conv.weight.detach().double().requires_grad_(True), conv.bias.detach().double().requires_grad_(True)
    ref = F.mish(F.conv2d(F.avg_pool2d(xd, 3, stride=2, padding=1), wd, bd, stride=2, padding=3)).reshape(B, -1)
    ref.backward(gout.double())
    assert (out.double() - ref).abs().max().item() < TOL
    for got, want, what in ((x.grad, xd.grad, "input"), (conv.weight.grad, wd.grad, "weight"), (conv.bias.grad, bd.grad, "bias")):
        scale = max(1.0, want.abs().max().item())
        assert (got.double() - want).abs().max().item() < 2e-5 * scale, (what, (got.double() - want).abs().max().item(), scale)


def test_net_forward_uses_dense_tail_and_matches_plain_module(fused):
    from Net.DQNNet import Net
    torch.manual_seed(5)
    net = Net(4, 10).cuda().eval()
    x = torch.randn(33, 4, 12, 12, device="cuda")
    q = net(x)
    loss = q.square().mean()
    loss.backward()
    g1 = [p.grad.clone() for p in net.parameters()]
    net.zero_grad()
    ref = net.double()._forward_plain(x.double())
    ref.square().mean().backward()
    assert (q.double() - ref).abs().max().item() < TOL
    for a, p in zip(g1, net.parameters()):
        assert (a.double() - p.grad).abs().max().item() < 2e-5 * max(1.0, p.grad.abs().max().item())


def test_presplit_weights_give_the_same_bits(fused):
    """tron_conv3x3_split_weights (all layers' weight images in one launch) + TRON_CONV_F16X3_PRESPLIT == the per-call
    split: same kernel, same operands."""
    torch.manual_seed(11)
    convs = [torch.nn.Conv2d(4, 32, 3, padding=1).cuda(), torch.nn.Conv2d(32, 64, 3, padding=1).cuda(),
             torch.nn.Conv2d(64, 64, 3, padding=1).cuda()]
    ws = fused.split_weights(convs)
    x = torch.randn(77, 4, 12, 12, device="cuda")
    for conv, w in zip(convs, ws):
        a = fused.conv3x3(x, conv, math="f16x3")
        b = fused.conv3x3(x, conv, math="f16x3", presplit=w)
        assert torch.equal(a, b)
        x = a
    from tron import _native as nat
    import ctypes as C
    one = (C.c_void_p * 1)(convs[0].weight.data_ptr())
    assert nat.lib().tron_conv3x3_split_weights(one, (C.c_int32 * 1)(4), (C.c_int32 * 1)(32), (C.c_void_p * 1)(None), 1, None) == nat.ERR_BAD_ARG
    assert nat.lib().tron_conv3x3_split_weights(one, (C.c_int32 * 1)(4), (C.c_int32 * 1)(32), (C.c_void_p * 1)(ws[0].data_ptr()), 9, None) == nat.ERR_BAD_ARG


@pytest.mark.parametrize("S,B", [(26, 3), (26, 130), (12, 5)])
def test_pool_s2_matches_avg_pool2d(fused, S, B):
    torch.manual_seed(S + B)
    x = torch.randn(B, 64, S, S, device="cuda")
    y = fused.pool_s2(x)
    ref = F.avg_pool2d(x.double(), 3, stride=2, padding=1)
    assert y.shape == ref.shape and (y.double() - ref).abs().max().item() < 1e-6


def test_net_act_in_eval_mode_uses_infer_and_agrees_with_the_module(fused):
    from Net.DQNNet import Net
    torch.manual_seed(21)
    net = Net(3, 10).cuda().eval()
    x = torch.randn(300, 3, 12, 12, device="cuda")
    with torch.no_grad():
        q = net(x)
    a = net.act(x)
    top = q.topk(2, dim=1).values
    clear = (top[:, 0] - top[:, 1]) > 1e-4                   # (the two paths agree to ~1e-6: near-ties may break either way)
    assert a.dtype == torch.int64 and clear.sum() > 250 and torch.equal(a[clear], q.argmax(1)[clear])


@pytest.mark.parametrize("S,B", [(26, 3), (26, 130), (34, 7), (12, 5)])
def test_pool_s2_backward_matches_autograd(fused, S, B):
    """Net/activations.py::_PoolS2 (tron_pool_s2 / tron_pool_s2_bwd: DQNNet.py:20,52 on the training path at 24x24 and
    32x32 boards) against autograd through float64 avg_pool2d."""
    from Net.activations import pool_s2, pool_s2_supported
    torch.manual_seed(S * B)
    pool = torch.nn.AvgPool2d(kernel_size=3, padding=1, stride=2)
    x = torch.randn(B, 64, S, S, device="cuda", requires_grad=True)
    assert pool_s2_supported(pool, x)
    y = pool_s2(pool, x)
    g = torch.randn_like(y)
    y.backward(g)
    xd = x.detach().double().requires_grad_(True)
    ref = F.avg_pool2d(xd, 3, stride=2, padding=1)
    ref.backward(g.double())
    assert (y.double() - ref).abs().max().item() < 1e-6 and (x.grad.double() - xd.grad).abs().max().item() < 1e-6


@pytest.mark.parametrize("cin", [3, 4])
def test_conv1_input_gradient_falls_back_to_the_library(fused, cin):
    """conv1's planes asked for their gradient: tron_conv3x3_dgrad has no cin 3 / 4 instantiation, so
    _ConvBiasMishHIP hands that one gradient to aten.convolution_backward instead of raising."""
    from Net.activations import conv_bias_mish
    torch.manual_seed(cin)
    conv = torch.nn.Conv2d(cin, 32, 3, padding=1).cuda()
    x = torch.randn(9, cin, 12, 12, device="cuda", requires_grad=True)
    out = conv_bias_mish(conv, x)
    assert type(out.grad_fn).__name__ == "_ConvBiasMishHIPBackward"
    g = torch.randn_like(out)
    out.backward(g)
    xd = x.detach().double().requires_grad_(True)
    F.mish(F.conv2d(xd, conv.weight.double(), conv.bias.double(), padding=1)).backward(g.double())
    assert (x.grad.double() - xd.grad).abs().max().item() < 2e-5 * max(1.0, xd.grad.abs().max().item())


@pytest.mark.parametrize("magnitude", [1.0, 1e-6])
@pytest.mark.parametrize("B", [1, 2, 5, 131, 300])
@pytest.mark.parametrize("cin,cout", [(32, 32), (32, 64), (64, 64)])
@pytest.mark.parametrize("S", [26, 34])
def test_conv3x3_wgrad_26_matches_float64(fused, S, cin, cout, B, magnitude):
    """tron_conv3x3_wgrad at 26x26 and 34x34 (24x24 / 32x32 boards: csrc/tron_conv_wgrad_rows.hip, image rows streamed through LDS,
    a 34-pixel row as two column halves) against the float64 weight gradient of F.conv2d; ragged batches (fewer images than
    workgroups, several per workgroup)."""
    torch.manual_seed(B * 17 + cin + cout + S)
    x = torch.randn(B, cin, S, S, device="cuda")
    gp = torch.randn(B, cout, S, S, device="cuda") * magnitude
    gp[0, 0, 0, 0] = 0.0
    assert fused.wgrad_supported(torch.empty(cout, cin, 3, 3, device="cuda"), S)
    wd = torch.zeros(cout, cin, 3, 3, dtype=torch.float64, device="cuda", requires_grad=True)
    F.conv2d(x.double(), wd, padding=1).backward(gp.double())
    ref = wd.grad
    blocks = gp.abs().reshape(B, -1).amax(1).contiguous()
    for absmax in (None, blocks):
        got = fused.conv3x3_wgrad(x, gp, absmax)
        err = (got.double() - ref).abs().max().item() / ref.abs().max().item()
        assert err < 2e-6, (err, absmax is None)
    assert torch.equal(fused.conv3x3_wgrad(x, gp, blocks), got)                    # deterministic
    # one-hot / asymmetric data: a swapped tap or channel cannot hide
    x = torch.zeros(B, cin, S, S, device="cuda")
    gp = torch.zeros(B, cout, S, S, device="cuda")
    x[:, 3, 0, S - 1] = 1.0
    x[:, cin - 1, S - 1, 0] = 2.0
    x[:, 7, 13, 13] = 3.0
    x[:, 11, 20, S // 2] = 5.0                                                     # (34: both sides of the seam between the column halves)
    x[:, 12, 21, S // 2 - 1] = 6.0
    gp[:, 5, 1, S - 2] = 1.0
    gp[:, cout - 2, S - 2, 1] = 4.0
    gp[:, 9, 13, 12] = 2.0
    gp[:, 3, 20, S // 2 - 1] = 7.0
    gp[:, 4, 21, S // 2] = 8.0
    gp[:, 6, 20, S // 2] = 1.0
    wd = torch.zeros(cout, cin, 3, 3, dtype=torch.float64, device="cuda", requires_grad=True)
    F.conv2d(x.double(), wd, padding=1).backward(gp.double())
    assert torch.equal(fused.conv3x3_wgrad(x, gp).double(), wd.grad)              # small integers: exact


@pytest.mark.parametrize("S,B", [(34, 1), (34, 6), (34, 67), (34, 300)])   # 300 x 4 bands: several groups per persistent workgroup
@pytest.mark.parametrize("cin,cout", [(32, 32), (32, 64), (64, 64), (64, 32), (3, 32), (4, 32)])
def test_conv3x3_side_34_matches_float64(fused, S, B, cin, cout):
    """32x32 boards (BASELINE config 5, the ACKTR nets' trunk): the split-f16 kernel with four row bands per image —
    forward with bias / residual / activation, and the input gradient."""
    if cin <= 4 and B > 67:
        pytest.skip("conv1: one size is enough")
    torch.manual_seed(S + B + cin + cout)
    conv = torch.nn.Conv2d(cin, cout, 3, padding=1).cuda()
    assert fused.supported(conv, 34)
    x = torch.randn(B, cin, S, S, device="cuda")
    res = torch.randn(B, cout, S, S, device="cuda")
    for r, act in ((res, True), (None, False)):
        got, pre = fused.conv3x3(x, conv, residual=r, act=act, want_pre=True)
        ref_pre, ref = _ref(x, conv, r, act)
        assert (pre.double() - ref_pre).abs().max().item() < TOL
        assert (got.double() - ref).abs().max().item() < TOL
    with torch.no_grad():
        conv.weight.copy_(torch.arange(conv.weight.numel(), device="cuda").reshape(conv.weight.shape).float() % 17 - 8)
    x = torch.zeros(B, cin, S, S, device="cuda")
    for row in (0, 7, 8, 15, 16, 23, 24, 33):                              # both sides of every band edge
        x[:, 1, row, (row * 5) % S] = 1.0 + row
    x[:, cin - 1, S - 1, 0] = 2.0
    assert torch.equal(fused.conv3x3(x, conv, act=False), F.conv2d(x, conv.weight, conv.bias, padding=1))   # small integers: exact
    if cin >= 32:
        for magnitude in (1.0, 1e-7):
            gp = torch.randn(B, cout, S, S, device="cuda") * magnitude
            xd = torch.zeros(B, cin, S, S, dtype=torch.float64, device="cuda", requires_grad=True)
            F.conv2d(xd, conv.weight.double(), padding=1).backward(gp.double())
            got = fused.conv3x3_dgrad(gp, conv.weight.detach(), gp.abs().amax().reshape(1))
            assert (got.double() - xd.grad).abs().max().item() / xd.grad.abs().max().item() < 2e-6


@pytest.mark.parametrize("S,B,cin,cout", [(12, 9, 3, 32), (12, 40, 32, 64), (26, 5, 64, 64), (34, 3, 4, 32), (34, 4, 32, 32), (34, 5, 64, 64),
                                          (14, 3, 32, 32), (12, 3, 32, 48)])
@pytest.mark.parametrize("bias", [True, False])
def test_conv3x3_module_matches_float64_conv2d(fused, S, B, cin, cout, bias):
    """Net/activations.py::Conv3x3 — the nn.Conv2d of the ACKTR nets (ACNet.py:97-116) with forward, input gradient and
    weight gradient on the hand-written kernels (the weight gradient on the library at 34x34); shapes they do not cover
    (side 14, 48 channels) run the library convolution."""
    from Net.activations import Conv3x3
    torch.manual_seed(S + B + cin + cout)
    m = Conv3x3(cin, cout, 3, padding=1, bias=bias).cuda()
    ref = torch.nn.Conv2d(cin, cout, 3, padding=1, bias=bias).cuda().double()
    ref.load_state_dict({k: v.double() for k, v in m.state_dict().items()})
    x = torch.randn(B, cin, S, S, device="cuda", requires_grad=True)
    xd = x.detach().double().requires_grad_(True)
    g = torch.randn(B, cout, S, S, device="cuda") * 1e-3
    y, yd = m(x), ref(xd)
    assert (y.double() - yd).abs().max().item() < TOL
    y.backward(g)
    yd.backward(g.double())
    for got, want in [(x.grad, xd.grad), (m.weight.grad, ref.weight.grad)] + ([(m.bias.grad, ref.bias.grad)] if bias else []):
        assert (got.double() - want).abs().max().item() / want.abs().max().item() < 1e-5
    # K-FAC's hooks sit on the module: they see the input and the output gradient as with the library convolution
    seen = {}
    m.register_forward_pre_hook(lambda mod, inp: seen.__setitem__("a", inp[0]))
    m.register_full_backward_hook(lambda mod, gi, go: seen.__setitem__("g", go[0]))
    x2 = x.detach().clone().requires_grad_(True)
    m(x2).backward(g)
    assert torch.equal(seen["a"], x2) and torch.equal(seen["g"], g)


@pytest.mark.parametrize("M,N,K,ta,tb", [(1, 64, 64, False, False), (300, 576, 2304, False, False), (4096, 2304, 576, False, True),
                                          (576, 2304, 4096, True, True), (576, 2304, 300, True, True), (129, 128, 1000, True, True),
                                          (4099, 64, 128, False, False),
                                          # k_gemm_wide (N % 192 == 0, K >= 1024, >= 32 tiles of 128 x 192) with ragged M: one, two and four K slices
                                          (12345, 576, 2304, False, False), (5000, 576, 2304, False, False), (3000, 576, 2304, False, False)])
@pytest.mark.parametrize("magnitude", [1.0, 1e-7])
def test_gemm_f16x3_matches_float64(fused, M, N, K, ta, tb, magnitude):
    """tron_gemm_f16x3 (the training head's dense products, Net/activations.py::_PoolConv7): every operand layout, ragged M,
    a reduction length that needs zero padding, a gradient-sized A operand with its device-side power-of-two scale; the 128 x 64 tile
    and the 128 x 192 tile with its K slices (csrc/tron_head.hip: wide_splitk)."""
    from Net.kfac import _pow2_scale
    torch.manual_seed(M + N + K)
    a = torch.randn((K, M) if ta else (M, K), device="cuda") * magnitude
    b = torch.randn((K, N) if tb else (N, K), device="cuda")
    bias = torch.randn(N, device="cuda") * magnitude
    got = fused.gemm_f16x3(a, b, bias, a_transposed=ta, b_transposed=tb, a_scale=_pow2_scale(a) if magnitude != 1.0 else None)
    want = (a.double().t() if ta else a.double()) @ (b.double() if tb else b.double().t()) + bias.double()
    assert got is not None and got.shape == want.shape
    assert (got.double() - want).abs().max().item() / want.abs().max().item() < 2e-6
    assert fused.gemm_f16x3(torch.randn(8, 100, device="cuda"), torch.randn(64, 100, device="cuda")) is None      # K % 64 != 0, not transposed
    assert fused.gemm_f16x3(torch.randn(8, 64, device="cuda"), torch.randn(48, 64, device="cuda")) is None        # N % 64 != 0


@pytest.mark.parametrize("magnitude", [1.0, 1e-6])
@pytest.mark.parametrize("B", [1, 5, 131, 1100])
@pytest.mark.parametrize("side,cin", [(26, 3), (26, 4), (34, 3), (34, 4)])
def test_conv1_wgrad_at_24x24_and_32x32_boards(fused, side, cin, B, magnitude):
    """conv1's weight gradient (3 or 4 planes -> 32) at 26x26 / 34x34 — plain f32 FMAs in tron_conv3x3_wgrad — against float64;
    1 100: more images than workgroups.  The planes are the observation encoding's values (DQNNet.py:33 on game.py's pop_up)."""
    torch.manual_seed(B + side + cin)
    vals = torch.tensor([0.0, 1.0, -1.0, 10.0, -10.0, 0.25], device="cuda")
    x = vals[torch.randint(0, 6, (B, cin, side, side), device="cuda")].contiguous()
    gp = torch.randn(B, 32, side, side, device="cuda") * magnitude
    assert fused.wgrad_supported(torch.empty(32, cin, 3, 3, device="cuda"), side)
    wd = torch.zeros(32, cin, 3, 3, dtype=torch.float64, device="cuda", requires_grad=True)
    F.conv2d(x.double(), wd, padding=1).backward(gp.double())
    got = fused.conv3x3_wgrad(x, gp)
    assert (got.double() - wd.grad).abs().max().item() / wd.grad.abs().max().item() < 3e-6
    assert torch.equal(fused.conv3x3_wgrad(x, gp), got)                            # deterministic
    x = torch.zeros(B, cin, side, side, device="cuda")                             # one-hot: a swapped tap or channel cannot hide
    gp = torch.zeros(B, 32, side, side, device="cuda")
    x[:, 0, 0, side - 1] = 1.0
    x[:, cin - 1, side - 1, 0] = 2.0
    x[:, 1, 13, 13] = 3.0
    gp[:, 5, 1, side - 2] = 1.0
    gp[:, 30, side - 2, 1] = 4.0
    gp[:, 9, 13, 12] = 2.0
    wd = torch.zeros(32, cin, 3, 3, dtype=torch.float64, device="cuda", requires_grad=True)
    F.conv2d(x.double(), wd, padding=1).backward(gp.double())
    assert torch.equal(fused.conv3x3_wgrad(x, gp).double(), wd.grad)              # small integers: exact
