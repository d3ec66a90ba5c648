"""csrc/tron_conv_ws.hip — the weight-stationary 3x3 convolutions on PX16 images (gradient-free forwards: the policy
and the DDQN target forwards, DDQN.py:90-110,129-142) — against a float64 torch reference of the same op
(DQNNet.py:33-50), tolerance 1e-5 (the north star's bound for Q-values)."""
import numpy as np
import pytest

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


def _codes(B, S, seed):
    gen = torch.Generator(device="cuda").manual_seed(seed)
    vals = torch.tensor([1, -1, -2, -3, 10, -10], dtype=torch.int8, device="cuda")
    return vals[torch.randint(0, 6, (B, S, S), device="cuda", generator=gen)]


def _to_px16(fused, x):
    """f32 [B, C, S, S] -> PX16 by the definition in include/tron_hip.h (host-side restatement for the tests)."""
    B, C, S, _ = x.shape
    s = (x.double() / 64.0)
    hi = s.to(torch.float16)
    lo = ((s - hi.double()) * 2048.0).to(torch.float16)
    img = torch.stack([hi, lo], 1)                                       # [B, 2, C, S, S]
    img = img.reshape(B, 2, C // 8, 8, S * S).permute(0, 1, 2, 4, 3).contiguous()   # [B, half, octet, pixel, 8]
    px = fused.PX16(B, C, S, x.device)
    px.buf.copy_(img.view(torch.uint8).reshape(-1))
    return px


@pytest.mark.parametrize("S,B,cin", [(12, 1, 3), (12, 777, 4), (26, 2, 3), (26, 130, 4), (34, 5, 3)])
def test_conv1_px16_matches_float64(fused, S, B, cin):
    """conv1 as a table sum over the codes (util.py:11-37 planes implied) == float64 conv of the pop_up planes."""
    from tron.vec import pop_up_planes
    torch.manual_seed(S + B + cin)
    conv = torch.nn.Conv2d(cin, 32, 3, padding=1).cuda()
    codes = _codes(B, S, S * B)
    planes = pop_up_planes(codes)
    if cin == 4:
        planes = torch.cat([planes, torch.full((B, 1, S, S), 5.0, device="cuda")], 1)
    ref = F.mish(F.conv2d(planes.double(), conv.weight.double(), conv.bias.double(), padding=1))
    got = fused.conv1_px16(codes, conv, 5.0).float()
    assert (got.double() - ref).abs().max().item() < TOL


@pytest.mark.parametrize("S,B", [(12, 1), (12, 2), (12, 7), (12, 260), (12, 1555), (26, 1), (26, 3), (26, 130), (26, 301)])
@pytest.mark.parametrize("cin,cout", [(32, 32), (32, 64), (64, 64)])
def test_conv_ws_matches_float64_reference(fused, S, B, cin, cout):
    """Every instantiation, ragged batches (items of two images, more items than workgroups), with / without the
    residual and the activation; PX16 output, f32 output and pre-activation all from one launch."""
    torch.manual_seed(S * 1000 + B + cin + cout)
    conv = torch.nn.Conv2d(cin, cout, 3, padding=1).cuda()
    x = torch.randn(B, cin, S, S, device="cuda")
    res = torch.randn(B, cout, S, S, device="cuda")
    xp, rp = _to_px16(fused, x), _to_px16(fused, res)
    assert (xp.float() - x).abs().max().item() < 1e-6                   # the format round-trips (2^-22 relative)
    w = fused.ws_split_weights([conv])[0]
    for r, rpx, act in ((res, rp, True), (None, None, True), (res, rp, False)):
        out, o32, pre = fused.conv_ws(xp, conv, w, residual=rpx, act=act, want_f32=True, want_pre=True)
        y = F.conv2d(x.double(), conv.weight.double(), conv.bias.double(), padding=1)
        if r is not None:
            y = y + r.double()
        ref = F.mish(y) if act else y
        assert (pre.double() - y).abs().max().item() < TOL
        assert (o32.double() - ref).abs().max().item() < TOL
        assert (out.float().double() - ref).abs().max().item() < TOL
    # asymmetric weights / one-hot inputs: a transposed tap or a swapped row/column cannot hide
    with torch.no_grad():
        conv.weight.copy_(torch.arange(conv.weight.numel(), device="cuda").reshape(conv.weight.shape).float() % 17 - 8)
        conv.bias.zero_()
    x = torch.zeros(B, cin, S, S, device="cuda")
    x[:, 1, 2, 3] = 1.0
    x[:, cin - 1, S - 1, 0] = 2.0
    x[:, 9, 0, S - 1] = -1.0
    w = fused.ws_split_weights([conv])[0]
    got = fused.conv_ws(_to_px16(fused, x), conv, w, act=False, want_px=False, want_f32=True)
    assert torch.equal(got, F.conv2d(x, conv.weight, None, padding=1))   # small integers: exact


@pytest.mark.parametrize("W,B,cin", [(10, 513, 3), (10, 64, 4), (24, 37, 3), (24, 200, 4)])
def test_trunk_px_matches_float64_module(fused, W, B, cin):
    """The whole chain conv1..conv6 from the codes against the float64 module's trunk."""
    from Net.DQNNet import Net
    from tron.vec import pop_up_planes
    torch.manual_seed(W + B)
    S = W + 2
    net = Net(cin, W).cuda()
    assert fused.ws_supported(net, S)
    codes = _codes(B, S, W * B)
    planes = pop_up_planes(codes)
    if cin == 4:
        planes = torch.cat([planes, torch.full((B, 1, S, S), 5.0, device="cuda")], 1)
    n64 = net.double()
    with torch.no_grad():
        x = F.mish(n64.conv1(planes.double()))
        idx = x
        x = F.mish(n64.conv2(x))
        x = F.mish(n64.conv3(x) + idx)
        x = F.mish(n64.conv4(x))
        idx = x
        x = F.mish(n64.conv5(x))
        ref = F.mish(n64.conv6(x) + idx)
    net.float()
    got = fused.trunk_px(net, codes, 5.0)
    assert (got.double() - ref).abs().max().item() < TOL
    assert (fused.trunk_px(net, codes, 5.0, want="px16").float() - got).abs().max().item() < 1e-6   # PX16 holds 2^-22 relative
    assert torch.equal(fused.trunk(net, codes, codes=True, plane4=5.0), got)         # the default gradient-free trunk


def test_conv_ws_bad_args(fused):
    from tron import _native as nat
    L = nat.lib()
    assert L.tron_conv3x3_ws_workspace(48, 32) == 0 and L.tron_conv3x3_ws_workspace(64, 64) == 4 * 18 * 2 * 1024
    assert L.tron_px16_bytes(3, 64, 12) == 3 * 64 * 144 * 4
    conv = torch.nn.Conv2d(64, 32, 3, padding=1).cuda()
    x = fused.PX16(2, 64, 12, torch.device("cuda"))
    w = fused.ws_split_weights([conv])[0]
    with pytest.raises(nat.TronNativeError):                             # 64 -> 32 has no instantiation
        fused.conv_ws(x, conv, w)
    with pytest.raises(TypeError):
        fused.conv_ws(fused.PX16(2, 32, 12, torch.device("cuda")), conv, w)
    a = (x.buf.data_ptr(), w.data_ptr(), None, None)
    assert L.tron_conv3x3_ws_fwd(*a, None, None, None, 2, 64, 64, 12, 1, None) == nat.ERR_BAD_ARG      # no output
    assert L.tron_conv3x3_ws_fwd(*a, x.buf.data_ptr(), None, None, 0, 64, 64, 12, 1, None) == 0        # empty batch
    assert L.tron_conv3x3_ws_fwd(*a, x.buf.data_ptr(), None, None, 2, 64, 64, 14, 1, None) == nat.ERR_UNSUPPORTED


@pytest.mark.parametrize("W,B", [(10, 1), (10, 777), (24, 3), (24, 140)])
def test_head_from_px16_equals_head_from_f32(fused, W, B):
    """tron_dqn_head_fwd_px16 (the pooling reads conv6's PX16 image) against the float64 reference of the tail
    (DQNNet.py:52-63) and against the f32-input head on the same values."""
    from Net.DQNNet import Net
    torch.manual_seed(W * B)
    S = W + 2
    net = Net(3, W).cuda()
    x = torch.randn(B, 64, S, S, device="cuda") * 1.5
    xp = _to_px16(fused, x)
    q_px, g_px = fused.head(net, xp, want_greedy=True)
    q_f = fused.head(net, xp.float())
    d = lambda t: t.double()
    y = F.avg_pool2d(x.double(), 3, stride=2, padding=1)
    y = F.mish(F.conv2d(y, d(net.conv7.weight), d(net.conv7.bias), stride=2, padding=3)).reshape(B, -1)
    y = F.mish(F.linear(y, d(net.fc1.weight), d(net.fc1.bias)))
    y = F.mish(F.linear(y, d(net.fc2.weight), d(net.fc2.bias)))
    ref = F.linear(F.mish(F.linear(y, d(net.actor1.weight), d(net.actor1.bias))), d(net.actor2.weight), d(net.actor2.bias))
    assert (q_px.double() - ref).abs().max().item() < TOL
    assert (q_px - q_f).abs().max().item() < 2e-6
    assert torch.equal(g_px.long(), q_px.argmax(1))


@pytest.mark.parametrize("W", [10, 24])
def test_infer_from_codes_takes_the_ws_chain(fused, W, monkeypatch):
    """Net.infer(codes) runs conv1..conv6 + head without an f32 activation tensor; Net.infer_path reports the path; the
    chunked layer kernels (TRON_CONV_WS=0) give the same Q to 1e-6."""
    from Net.DQNNet import Net
    torch.manual_seed(W)
    S = W + 2
    net = Net(3, W).cuda()
    codes = _codes(300, S, W)
    assert net.infer_path(codes, codes=True) == "ws-chain"
    assert net.infer_path(torch.zeros(2, 3, S, S, device="cuda")) == "layer-kernels"
    assert net.infer_path(torch.zeros(2, 3, 14, 14, device="cuda")) == "module"
    q = net.infer(codes, codes=True)
    monkeypatch.setattr(fused, "use_ws", False)
    assert net.infer_path(codes, codes=True) == "layer-kernels"
    q_old = net.infer(codes, codes=True)
    assert (q - q_old).abs().max().item() < 2e-6


@pytest.mark.parametrize("B", [1, 2, 255, 256, 257, 777, 4096, 8192 + 3])
def test_conv6_with_the_pooling_in_one_launch_has_the_bits_of_two(fused, B, monkeypatch):
    """tron_conv3x3_ws_fwd_pool12 (conv6's output stays in LDS, DQNNet.py:48-52) against tron_conv3x3_ws_fwd followed by the head's own
    pooling: the pooled rows, Q and the greedy action bit for bit; and against float64 through Net.infer.  Batches below, at and
    above one image per workgroup (256 CUs), and several images per workgroup with a ragged last round."""
    from Net.DQNNet import Net
    from tron import _native as nat
    torch.manual_seed(B)
    net = Net(3, 10).cuda()
    x = _to_px16(fused, torch.randn(B, 64, 12, 12, device="cuda") * 1.5)
    r = _to_px16(fused, torch.randn(B, 64, 12, 12, device="cuda"))
    w = fused.ws_split_weights([net.conv6])[0]
    pooled = fused.conv_ws_pool12(x, net.conv6, w, r)
    assert pooled.buf.numel() == nat.lib().tron_pooled12_bytes(B) == 2 * ((B * 2304 * 2 + 255) // 256 * 256)
    y = fused.conv_ws(x, net.conv6, w, residual=r)                        # the PX16 image the two-launch path writes
    # the head's pooling of that image, by the definition: the window's values hi + lo 2^-11 added in tap order, / 9, split again
    img = y.buf.view(torch.float16).reshape(B, 2, 8, 12, 12, 8).float()   # [b, half, octet, y, x, channel]
    pad = torch.nn.functional.pad(img[:, 0] + img[:, 1] * (1.0 / 2048.0), (0, 0, 1, 1, 1, 1))
    acc = torch.zeros(B, 8, 6, 6, 8, device="cuda")
    for t in range(9):
        acc = acc + pad[:, :, t // 3:t // 3 + 12:2, t % 3:t % 3 + 12:2, :]
    s = acc * (1.0 / 9.0)
    hi = s.to(torch.float16)
    lo = ((s - hi.float()) * 2048.0).to(torch.float16)
    half = (B * 2304 * 2 + 255) // 256 * 256
    got = pooled.buf.view(torch.float16)
    assert torch.equal(got[:B * 2304].reshape(B, 8, 6, 6, 8), hi)
    assert torch.equal(got[half // 2:half // 2 + B * 2304].reshape(B, 8, 6, 6, 8), lo)
    q1, g1 = fused.head(net, pooled, want_greedy=True)
    q2, g2 = fused.head(net, y, want_greedy=True)
    assert torch.equal(q1, q2) and torch.equal(g1, g2)


@pytest.mark.parametrize("B", [300, 5000])
def test_infer_at_12x12_pools_inside_conv6(fused, B, monkeypatch):
    """Net.infer(codes) at 12x12 takes the one-launch conv6 + pooling; TRON_POOL_FUSED=0 (two launches) gives the same bits."""
    from Net.DQNNet import Net
    torch.manual_seed(B)
    net = Net(3, 10).cuda()
    codes = _codes(B, 12, B)
    calls = []
    real = fused.conv_ws_pool12
    monkeypatch.setattr(fused, "conv_ws_pool12", lambda *a: (calls.append(1), real(*a))[1])
    q, g = net.infer(codes, codes=True), net.infer(codes, codes=True, greedy=True)
    assert len(calls) == 2
    monkeypatch.setattr(fused, "use_pool_fused", False)
    assert torch.equal(net.infer(codes, codes=True), q) and torch.equal(net.infer(codes, codes=True, greedy=True), g)
    assert len(calls) == 2


def test_pool12_bad_args(fused):
    from tron import _native as nat
    L = nat.lib()
    assert L.tron_pooled12_bytes(-1) == 0 and L.tron_pooled12_bytes(0) == 0 and L.tron_pooled12_bytes(1) == 2 * 4608
    buf = torch.zeros(1 << 16, dtype=torch.uint8, device="cuda")
    p = buf.data_ptr()
    assert L.tron_conv3x3_ws_fwd_pool12(p, p, p, None, p, 1, None) == nat.ERR_BAD_ARG      # conv6 always has its residual
    assert L.tron_conv3x3_ws_fwd_pool12(p + 8, p, p, p, p, 1, None) == nat.ERR_BAD_ARG
    assert L.tron_conv3x3_ws_fwd_pool12(p, p, p, p, p, 0, None) == nat.OK
    net_w = [buf.data_ptr()] * 10
    assert L.tron_dqn_head_fwd_pooled(p, 1, 26, *net_w, p, p, None, None) == nat.ERR_UNSUPPORTED
