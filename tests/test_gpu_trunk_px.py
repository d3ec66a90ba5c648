"""The learner's trunk on the weight-stationary design (csrc/tron_conv_ws_train.hip; DDQN.py:115-151 on DQNNet.py:33-50), every
piece through the C ABI against float64 autograd of the same expression:
  * the training forward (output + pre-activation as PX16 images),
  * the gradient chain's entry (g * mish'(z) as a gradient image, bias sums),
  * the input gradient as the weight-stationary kernel on rotated weights with the fused activation backward,
  * the weight gradient straight from two PX16 images (transposed LDS reads),
  * the whole node (`Net/activations.py::_TrunkPX`) against the float64 module and against the previous node."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")
F = torch.nn.functional


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")


@pytest.fixture(scope="module")
def fused():
    from Net import fused
    return fused


def _codes(B, S, seed):
    g = torch.Generator(device="cuda").manual_seed(seed)
    vals = torch.tensor([1, 1, 1, -1, -2, -3, 10, -10], dtype=torch.int8, device="cuda")
    return vals[torch.randint(0, 8, (B, S, S), device="cuda", generator=g)].contiguous()


def _to_px(fused, x, scale=None):
    """f32 [B, C, S, S] -> a PX16 image (host-side packing: test infrastructure) of x * scale."""
    B, C, S, _ = x.shape
    v = (x.double() * (1.0 if scale is None else scale) / 64.0)
    hi = v.to(torch.float16)
    lo = ((v - hi.double()) * 2048.0).to(torch.float16)
    def lay(t):                                                      # [B, C, S, S] -> [B][octet][pixel][8 channels]
        return t.reshape(B, C // 8, 8, S * S).permute(0, 1, 3, 2).contiguous()
    img = torch.stack([lay(hi), lay(lo)], 1).contiguous()            # [B][hi | lo][octet][pixel][8]
    px = fused.PX16(B, C, S, x.device)
    px.buf.copy_(img.view(torch.uint8).reshape(-1))
    return px


def _grad_px(fused, g):
    """A gradient image of g at the scale tron_absmax_pow2(14) would give (test infrastructure)."""
    B, C, S, _ = g.shape
    m = g.abs().max().item()
    s = 2.0 ** (13 - int(np.floor(np.log2(m)))) if m > 0 else 1.0       # max |g| s in [2^13, 2^14)
    px = _to_px(fused, g, s)
    out = fused.GradPX(B, C, S, g.device)
    out.buf = px.buf
    out.info = torch.zeros(68, dtype=torch.float32, device=g.device)
    out.info[0], out.info[1] = s, 1.0 / s
    out.info[4:4 + C] = g.abs().amax((0, 2, 3))
    return out


@pytest.mark.parametrize("S", [12, 26, 34])
@pytest.mark.parametrize("cin,cout,res", [(32, 32, False), (32, 32, True), (32, 64, False), (64, 64, False), (64, 64, True)])
def test_training_forward_keeps_the_pre_activation(fused, S, cin, cout, res):
    torch.manual_seed(cin + cout + S)
    B = 37 if S == 12 else 11
    conv = torch.nn.Conv2d(cin, cout, 3, padding=1).cuda()
    x = torch.randn(B, cin, S, S, device="cuda") * 1.5
    r = torch.randn(B, cout, S, S, device="cuda") if res else None
    frag = fused._split_jobs([conv.weight], "tron_conv3x3_ws_split_weights", False)[0]
    xp, rp = _to_px(fused, x), (None if r is None else _to_px(fused, r))
    a, z = fused.conv_ws_train(xp, cout, frag, conv.bias, residual=rp)
    a32, z2 = fused.conv_ws_train(xp, cout, frag, conv.bias, residual=rp, want_f32=True)
    zref = F.conv2d(x.double(), conv.weight.double(), conv.bias.double(), padding=1) + (0 if r is None else r.double())
    aref = F.mish(zref)
    assert (z.float().double() - zref).abs().max().item() < 1e-5
    assert (a.float().double() - aref).abs().max().item() < 1e-5
    assert (a32.double() - aref).abs().max().item() < 1e-5 and torch.equal(z.buf, z2.buf)
    if S != 34:                                     # the gradient-free chain's kernel (12x12 / 26x26 boards) computes the same output bits
        assert torch.equal(fused.conv_ws(xp, conv, frag, residual=rp).buf, a.buf)


@pytest.mark.parametrize("S,cin", [(12, 3), (26, 4)])
def test_conv1_training_forward(fused, S, cin):
    torch.manual_seed(S)
    B = 29
    conv = torch.nn.Conv2d(cin, 32, 3, padding=1).cuda()
    codes = _codes(B, S, 5)
    a, z = fused.conv1_px16_train(codes, conv.weight, conv.bias, 5.0)
    from tron.vec import pop_up_planes
    planes = pop_up_planes(codes).double()
    if cin == 4:
        planes = torch.cat([planes, torch.full_like(planes[:, :1], 5.0)], 1)
    zref = F.conv2d(planes, conv.weight.double(), conv.bias.double(), padding=1)
    assert (z.float().double() - zref).abs().max().item() < 2e-5
    assert (a.float().double() - F.mish(zref)).abs().max().item() < 2e-5
    assert torch.equal(a.buf, fused.conv1_px16(codes, conv, 5.0).buf)


@pytest.mark.parametrize("S,mag,B", [(12, 1.0, 19), (26, 1e-6, 19), (26, 1.0, 19), (12, 1.0, 16 * 1024 + 21), (26, 1.0, 4 * 1024 + 5)])
def test_gradient_image_entry_from_the_pooled_gradient(fused, S, mag, B):
    """tron_px16_grad_from_pooled: AvgPool2d(3, 2, 1)'s backward + mish'(z) + bias sums in one pass, from pooled planes (12x12) and
    from the channels-last pooled gradient (26x26), against float64 autograd of mish -> avg_pool."""
    from tron import _native as nat
    L = nat.lib()
    torch.manual_seed(S)
    C, PS = 64, S // 2                                # (the kernel walks 16 / 4 images at a time, <= 1024 workgroups per octet: ragged and looping batches)
    z = torch.randn(B, C, S, S, device="cuda") * 2.5
    gpool = torch.randn(B, C, PS, PS, device="cuda") * mag
    zd = z.double().requires_grad_(True)
    F.avg_pool2d(F.mish(zd), 3, stride=2, padding=1).backward(gpool.double())
    src = gpool.permute(0, 2, 3, 1).contiguous() if S == 26 else gpool.contiguous()
    sc4 = torch.zeros(4, device="cuda")
    out = fused.GradPX(B, C, S, "cuda")
    gb = torch.empty(C, device="cuda")
    ws = torch.empty(int(L.tron_px16_grad_workspace(B, C)), dtype=torch.uint8, device="cuda")
    nat.check(L.tron_absmax_pow2(nat.ptr(src), src.numel(), 15, nat.ptr(sc4), nat.stream_ptr()))
    nat.check(L.tron_px16_grad_from_pooled(nat.ptr(src), int(S == 26), nat.ptr(_to_px(fused, z).buf), nat.ptr(sc4), B, C, S, nat.ptr(out.buf),
                                           nat.ptr(out.info), nat.ptr(gb), nat.ptr(ws), nat.stream_ptr()))
    scale = zd.grad.abs().max().item()
    assert (out.float().double() - zd.grad).abs().max().item() < 2e-6 * scale
    sums = zd.grad.sum((0, 2, 3))
    assert (gb.double() - sums).abs().max().item() < 1e-5 * sums.abs().max().item() + 1e-6 * scale
    info = out.info.cpu().numpy()
    assert abs(info[4:4 + C].max() - scale) <= 2e-6 * scale and info[4:4 + C].max() * info[0] < 2.0 ** 15


@pytest.mark.parametrize("S,C,mag", [(12, 64, 1.0), (26, 64, 1e-6), (12, 32, 3e-8), (34, 64, 1.0)])
def test_gradient_image_entry(fused, S, C, mag):
    torch.manual_seed(S + C)
    B = 23
    z = torch.randn(B, C, S, S, device="cuda") * 3.0
    z[0, 0, 0, :4] = torch.tensor([25.0, -30.0, 0.0, 60.0], device="cuda")   # both tails of the activation
    g = torch.randn(B, C, S, S, device="cuda") * mag
    gp, gb = fused.grad_px_from_f32(g, _to_px(fused, z))
    zd = z.double().requires_grad_(True)
    F.mish(zd).backward(g.double())
    scale = zd.grad.abs().max().item()
    assert (gp.float().double() - zd.grad).abs().max().item() < 2e-6 * scale
    assert (gb.double() - zd.grad.sum((0, 2, 3))).abs().max().item() < 1e-5 * zd.grad.sum((0, 2, 3)).abs().max().item() + 1e-6 * scale
    info = gp.info.cpu().numpy()
    assert info[0] * info[1] == 1.0 and abs(info[4:4 + C].max() - scale) <= 2e-6 * scale
    assert (info[4:4 + C] >= zd.grad.abs().amax((0, 2, 3)).cpu().numpy() * (1 - 1e-5)).all()      # (an entry bounds its channel: this kernel keeps one maximum per octet)
    assert 2.0 ** 12 <= info[4:4 + C].max() * info[0] < 2.0 ** 15


DG = [(32, 32, False), (32, 32, True), (32, 64, False), (64, 64, False), (64, 64, True)]


@pytest.mark.parametrize("S", [12, 26, 34])
@pytest.mark.parametrize("cin,cout,extra", DG)
@pytest.mark.parametrize("mag", [1.0, 1e-6])
def test_input_gradient_with_fused_activation_backward(fused, S, cin, cout, extra, mag):
    """(conv^T(g, W) + extra) * mish'(z_below), its column sums and its f32 copy, for every layer shape of the trunk, with the
    gradient at unit size and at a mean-reduced loss's 1e-6."""
    torch.manual_seed(cin * 3 + cout + S)
    B = 21 if S == 12 else 7
    W = torch.randn(cout, cin, 3, 3, device="cuda") * 0.08
    g = torch.randn(B, cout, S, S, device="cuda") * mag
    zb = torch.randn(B, cin, S, S, device="cuda") * 2.0
    ex = torch.randn(B, cin, S, S, device="cuda") * mag * 3.0 if extra else None
    rot, wn = fused._split_jobs([W], "tron_conv3x3_ws_split_weights_bwd", True)
    assert abs(wn.item() - W.abs().sum((0, 2, 3)).max().item()) < 1e-4 * wn.item()
    out, o32, gb = fused.conv_ws_dgrad(_grad_px(fused, g), cin, rot[0], wn[0:1], _to_px(fused, zb),
                                       extra=None if ex is None else _grad_px(fused, ex), want_f32=True)
    xd = torch.zeros(B, cin, S, S, device="cuda", dtype=torch.float64, requires_grad=True)
    zd = zb.double()
    # reference: d/dx of <conv(x), g> is conv^T(g); then + extra, times mish'(z)
    (F.conv2d(xd, W.double(), padding=1) * g.double()).sum().backward()
    zz = zd.clone().requires_grad_(True)
    F.mish(zz).backward(xd.grad + (0 if ex is None else ex.double()))
    ref = zz.grad
    scale = ref.abs().max().item()
    assert (out.float().double() - ref).abs().max().item() < 4e-6 * scale
    assert (o32.double() - ref).abs().max().item() < 4e-6 * scale
    sums = ref.sum((0, 2, 3))
    assert (gb.double() - sums).abs().max().item() < 2e-5 * sums.abs().max().item() + 2e-6 * scale
    info = out.info.cpu().numpy()
    top = info[4:4 + cin].max()
    assert abs(top - scale) < 1e-5 * scale and top * info[0] < 2.0 ** 15 and top * info[0] >= 2.0 ** 5
    # deterministic
    out2, _, gb2 = fused.conv_ws_dgrad(_grad_px(fused, g), cin, rot[0], wn[0:1], _to_px(fused, zb),
                                       extra=None if ex is None else _grad_px(fused, ex))
    assert torch.equal(out.buf, out2.buf) and torch.equal(gb, gb2)


@pytest.mark.parametrize("S", [12, 26, 34])
@pytest.mark.parametrize("cin,cout", [(32, 32), (32, 64), (64, 64)])
@pytest.mark.parametrize("B,mag", [(37, 1.0), (6, 1e-6), (1, 1.0)])
def test_weight_gradient_from_px16_images(fused, S, cin, cout, B, mag):
    """dW = sum over images and pixels of g x shifted input, from the two PX16 images; odd batches (a last stack with one image
    at 12x12), 1e-6-sized gradients, and one-hot operands (exact: every tap's halo, both band edges, the image separator)."""
    torch.manual_seed(cin + cout + S + B)
    x = torch.randn(B, cin, S, S, device="cuda") * 1.5
    g = torch.randn(B, cout, S, S, device="cuda") * mag
    gw = fused.conv3x3_wgrad_px(_to_px(fused, x), _grad_px(fused, g))
    wd = torch.zeros(cout, cin, 3, 3, device="cuda", dtype=torch.float64, requires_grad=True)
    (F.conv2d(x.double(), wd, padding=1) * g.double()).sum().backward()
    scale = wd.grad.abs().max().item()
    assert (gw.double() - wd.grad).abs().max().item() < 3e-6 * scale
    assert torch.equal(gw, fused.conv3x3_wgrad_px(_to_px(fused, x), _grad_px(fused, g)))
    if mag == 1.0 and B > 1:
        # one-hot: a single unit gradient pixel against a single unit input pixel lights exactly one tap
        rs = np.random.RandomState(S + cin)
        for _ in range(6):
            b, co, ci = rs.randint(B), rs.randint(cout), rs.randint(cin)
            y, xx = rs.choice([0, 5, 7, 8, S - 1]), rs.choice([0, 3, S - 1])
            ky, kx = rs.randint(3), rs.randint(3)
            iy, ix = y + ky - 1, xx + kx - 1
            x1 = torch.zeros(B, cin, S, S, device="cuda")
            g1 = torch.zeros(B, cout, S, S, device="cuda")
            g1[b, co, y, xx] = 1.0
            if 0 <= iy < S and 0 <= ix < S:
                x1[b, ci, iy, ix] = 2.0
            x1[(b + 1) % B, ci, (iy + 1) % S, ix % S] = 7.0               # another image must not leak across the separator row
            got = fused.conv3x3_wgrad_px(_to_px(fused, x1), _grad_px(fused, g1))
            want = torch.zeros(cout, cin, 3, 3, device="cuda")
            if 0 <= iy < S and 0 <= ix < S:
                want[co, ci, ky, kx] = 2.0
            assert torch.equal(got, want), (b, co, ci, y, xx, ky, kx)


@pytest.mark.parametrize("S", [12, 26])
@pytest.mark.parametrize("cin", [3, 4])
@pytest.mark.parametrize("B,mag", [(37, 1.0), (600, 1e-6), (1, 1.0)])
def test_conv1_weight_gradient_from_the_codes(fused, S, cin, B, mag):
    """conv1's weight gradient (DQNNet.py:10,34 in DDQN.py:148) from the int8 codes and the gradient image against a float64
    convolution backward on util.pop_up's planes (+ the constant fourth plane): batches smaller and larger than the grid,
    1e-6-sized gradients, and one-hot cases (a single unit gradient pixel beside a single head / wall cell: exactly the
    taps that see it, at image corners, row ends and both bands of a 26x26 board)."""
    from tron.vec import pop_up_planes
    torch.manual_seed(S + cin + B)
    codes = _codes(B, S, S + B)
    plane4 = 0.37
    planes = pop_up_planes(codes)
    if cin == 4:
        planes = torch.cat([planes, torch.full_like(planes[:, :1], plane4)], 1)
    g = torch.randn(B, 32, S, S, device="cuda") * mag
    gw = fused.conv1_wgrad_px(codes, _grad_px(fused, g), cin, plane4)
    wd = torch.zeros(32, cin, 3, 3, device="cuda", dtype=torch.float64, requires_grad=True)
    (F.conv2d(planes.double(), wd, padding=1) * g.double()).sum().backward()
    assert (gw.double() - wd.grad).abs().max().item() < 3e-6 * wd.grad.abs().max().item()
    assert torch.equal(gw, fused.conv1_wgrad_px(codes, _grad_px(fused, g), cin, plane4))
    if mag == 1.0 and B > 1:
        rs = np.random.RandomState(S + cin)
        for code, plane, val in ((-1, 0, 1.0), (10, 1, 10.0), (-2, 1, 1.0), (-10, 2, 10.0), (-3, 2, 1.0)):
            b, co = rs.randint(B), rs.randint(32)
            y, xx = rs.choice([0, S // 2 - 1, S // 2, S - 1]), rs.choice([0, 3, S - 1])
            ky, kx = rs.randint(3), rs.randint(3)
            iy, ix = y + ky - 1, xx + kx - 1
            c1 = torch.ones(B, S, S, dtype=torch.int8, device="cuda")       # EMPTY everywhere: no plane of the first three is set
            if 0 <= iy < S and 0 <= ix < S:
                c1[b, iy, ix] = code
            c1[(b + 1) % B, (iy + 1) % S, ix % S] = code                    # another image must not leak in
            g1 = torch.zeros(B, 32, S, S, device="cuda")
            g1[b, co, y, xx] = 1.0
            got = fused.conv1_wgrad_px(c1, _grad_px(fused, g1), 3, 0.0)
            want = torch.zeros(32, 3, 3, 3, device="cuda")
            if 0 <= iy < S and 0 <= ix < S:
                want[co, plane, ky, kx] = val
            assert torch.equal(got, want), (code, b, co, y, xx, ky, kx)


@pytest.mark.parametrize("body", ["1", "0"])
@pytest.mark.parametrize("W,B", [(10, 64), (24, 12), (10, 1), (24, 3)])
def test_trunk_node_matches_float64_and_the_previous_node(fused, W, B, body, monkeypatch):
    """Net.forward_codes -> `_BodyPX` (conv1 .. conv7 as one node; TRON_BODY_PX=0: `_TrunkPX`, conv1 .. conv6): output and every
    parameter gradient against the float64 module, and against the layer-kernel node they replace (TRON_TRUNK_PX=0)."""
    import copy
    from Net.DQNNet import Net
    monkeypatch.setenv("TRON_BODY_PX", body)
    torch.manual_seed(W)
    net = Net(3, W).cuda()
    net.dropout.p = 0.0
    codes = _codes(B, W + 2, 3)
    up = torch.randn(B, 4, device="cuda") * (1.0 / B)
    q = net.forward_codes(codes)
    (q * up).sum().backward()
    from tron.vec import pop_up_planes
    n64 = copy.deepcopy(net).double()
    for p in n64.parameters():
        p.grad = None
    q64 = n64._forward_plain(pop_up_planes(codes).double())
    (q64 * up.double()).sum().backward()
    assert (q.double() - q64).abs().max().item() < 1e-5
    for (n, p), p64 in zip(net.named_parameters(), n64.parameters()):
        scale = max(p64.grad.abs().max().item(), 1e-12)
        err = (p.grad.double() - p64.grad).abs().max().item()
        assert err < 5e-5 * scale, (n, err, scale)
    g_px = [p.grad.clone() for p in net.parameters()]
    monkeypatch.setattr(fused, "use_trunk_px", False)
    net.zero_grad(set_to_none=True)
    q2 = net.forward_codes(codes)
    (q2 * up).sum().backward()
    assert (q - q2).abs().max().item() < 1e-5
    for (n, p), gp in zip(net.named_parameters(), g_px):
        scale = max(p.grad.abs().max().item(), 1e-12)
        assert (p.grad - gp).abs().max().item() < 1e-4 * scale, n


def test_trunk_px_is_the_path_the_learner_takes(fused):
    from Net.DQNNet import Net
    from Net.activations import trunk_px_supported
    net = Net(3, 10).cuda()
    assert trunk_px_supported(net, _codes(4, 12, 1)) and not trunk_px_supported(net, torch.zeros(4, 3, 12, 12, device="cuda"))
    q = net.forward_codes(_codes(4, 12, 1))
    names = set()
    stack = [q.grad_fn]
    while stack:
        f = stack.pop()
        if f is None:
            continue
        names.add(type(f).__name__)
        stack += [n for n, _ in f.next_functions]
    assert "_BodyPXBackward" in names, names


@pytest.mark.parametrize("B", [1, 255, 257, 1000, 4096])
def test_training_conv6_with_the_pooling_in_one_launch(fused, B):
    """tron_conv3x3_ws_train_fwd_pool12 (the learner's conv6 + AvgPool2d(3, 2, 1), DQNNet.py:48-52, conv6's output left in LDS)
    against tron_conv3x3_ws_train_fwd followed by tron_pool12_px16: the pooled planes and the pre-activation image bit for bit."""
    from tron import _native as nat
    torch.manual_seed(B)
    x = _to_px(fused, torch.randn(B, 64, 12, 12, device="cuda") * 1.5)
    r = _to_px(fused, torch.randn(B, 64, 12, 12, device="cuda"))
    w = torch.randn(64, 64, 3, 3, device="cuda") * 0.05
    bias = torch.randn(64, device="cuda") * 0.1
    frag = fused._split_jobs([w], "tron_conv3x3_ws_split_weights", False)[0]
    pooled, z = fused.conv_ws_train_pool12(x, frag, bias, r)
    a6, z2 = fused.conv_ws_train(x, 64, frag, bias, residual=r)
    ref = torch.empty(B, 64 * 36, dtype=torch.float32, device="cuda")
    nat.check(nat.lib().tron_pool12_px16(nat.ptr(a6.buf), nat.ptr(ref), B, nat.stream_ptr()), "tron_pool12_px16")
    assert torch.equal(z.buf, z2.buf)
    assert torch.equal(pooled, ref)
    want = torch.nn.functional.avg_pool2d(a6.float().double(), 3, stride=2, padding=1).reshape(B, -1)
    assert (pooled.double() - want).abs().max().item() < 1e-5


def test_learner_body_pools_inside_conv6(fused, monkeypatch):
    """`_BodyPX` at 12x12 takes the one-launch conv6 + pooling; with two launches (TRON_POOL_FUSED=0) the output and every
    gradient have the same bits."""
    from Net.DQNNet import Net
    torch.manual_seed(5)
    net = Net(3, 10).cuda()
    net.dropout.p = 0.0
    codes = _codes(300, 12, 9)
    up = torch.randn(300, 4, device="cuda")
    calls = []
    real = fused.conv_ws_train_pool12
    monkeypatch.setattr(fused, "conv_ws_train_pool12", lambda *a: (calls.append(1), real(*a))[1])
    q = net.forward_codes(codes)
    (q * up).sum().backward()
    assert len(calls) == 1
    grads = [p.grad.clone() for p in net.parameters()]
    monkeypatch.setattr(fused, "use_pool_fused_train", False)
    net.zero_grad(set_to_none=True)
    q2 = net.forward_codes(codes)
    (q2 * up).sum().backward()
    assert len(calls) == 1 and torch.equal(q, q2)
    for (n, p), g in zip(net.named_parameters(), grads):
        assert torch.equal(p.grad, g), n
