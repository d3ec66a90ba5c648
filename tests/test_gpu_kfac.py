"""GPU tests of the K-FAC patch-extraction kernel (csrc/tron_kfac.hip) behind Net/kfac.py."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")


SHAPES = [  # B, C, H, W, k, pad, stride
    (5, 3, 12, 12, 3, 1, 1),      # conv1 at 10x10
    (3, 32, 26, 26, 3, 1, 1),     # conv2/3 at 24x24
    (2, 64, 34, 34, 3, 1, 1),     # conv5/6 at 32x32
    (4, 64, 17, 17, 7, 3, 2),     # conv7 after the pool at 32x32
    (7, 5, 9, 11, 3, 0, 1),       # no padding, odd sizes, non-square
    (6, 8, 13, 13, 3, 1, 2),      # strided 3x3
    (1, 1, 3, 3, 3, 1, 1),
]


@pytest.mark.parametrize("B,C,H,W,k,pad,stride", SHAPES)
def test_extract_patches_equals_unfold(B, C, H, W, k, pad, stride):
    """Pure data movement: bit-identical to F.unfold(...).transpose(1, 2).reshape(-1, C*k*k)."""
    import torch.nn.functional as F
    from Net.kfac import extract_patches
    torch.manual_seed(B * 100 + C)
    x = torch.randn(B, C, H, W, device="cuda")
    got = extract_patches(x, (k, k), (pad, pad), (stride, stride))
    cols = F.unfold(x, (k, k), padding=pad, stride=stride)
    want = cols.transpose(1, 2).reshape(-1, cols.size(1))
    assert got.shape == want.shape and torch.equal(got, want)
    # a non-contiguous view (a micro-batch slice of a bigger rollout tensor) works too
    big = torch.randn(B + 2, C, H, W, device="cuda")
    assert torch.equal(extract_patches(big[1:-1], (k, k), (pad, pad), (stride, stride)),
                       F.unfold(big[1:-1], (k, k), padding=pad, stride=stride).transpose(1, 2).reshape(-1, C * k * k))


def test_extract_patches_unsupported_falls_back_to_unfold():
    """Rows that do not fit 64 KB of LDS are refused by the kernel; the helper then uses F.unfold."""
    import torch.nn.functional as F
    import tron._native as nat
    from Net.kfac import extract_patches
    x = torch.randn(1, 256, 8, 40, device="cuda")
    out = torch.empty(8 * 40, 256 * 9, device="cuda")
    rc = nat.lib().tron_extract_patches(nat.ptr(x), 1, 256, 8, 40, 3, 3, 1, 1, nat.ptr(out), nat.stream_ptr())
    assert rc == nat.ERR_UNSUPPORTED
    want = F.unfold(x, (3, 3), padding=1, stride=1).transpose(1, 2).reshape(-1, 256 * 9)
    assert torch.equal(extract_patches(x, (3, 3), (1, 1), (1, 1)), want)
    assert nat.lib().tron_extract_patches(None, 1, 1, 3, 3, 3, 3, 1, 1, nat.ptr(out), nat.stream_ptr()) == nat.ERR_BAD_ARG
    assert nat.lib().tron_extract_patches(nat.ptr(x), 1, 256, 8, 40, 9, 9, 0, 1, nat.ptr(out), nat.stream_ptr()) == nat.ERR_BAD_ARG


@pytest.mark.parametrize("C,side,k,pad,stride", [(32, 12, 3, 1, 1), (64, 9, 7, 3, 2)])
def test_cov_inputs_gpu_matches_cpu(C, side, k, pad, stride):
    """The A-factor from the kernel + one GEMM == the reference-ordered CPU computation (kfac.py:41-58),
    also when the batch arrives in micro-batches."""
    import torch.nn as nn
    from Net.kfac import cov_inputs
    torch.manual_seed(1)
    conv = nn.Conv2d(C, 8, k, padding=pad, stride=stride)
    a = torch.randn(24, C, side, side)
    want = cov_inputs(a, conv)
    got = cov_inputs(a.cuda(), conv)
    assert torch.allclose(got.cpu(), want, rtol=1e-4, atol=1e-6)
    parts = sum(cov_inputs(a[i:i + 8].cuda(), conv, 24) for i in range(0, 24, 8))
    assert torch.allclose(parts.cpu(), want, rtol=1e-4, atol=1e-6)


@pytest.mark.parametrize("B,C,S", [(1, 32, 12), (7, 64, 12), (300, 32, 34), (33, 64, 34), (5000, 32, 12), (9, 64, 26), (4, 48, 12), (4, 64, 9)])
def test_gradient_factor_from_nchw_matches_float64(B, C, S):
    """tron_kfac_patch_gram with a 1x1 geometry — a convolution's gradient factor — runs k_gram_nchw for 32 / 64 channels and
    positions % 4 == 0 (the last two shapes take the general path): float64 reference, symmetric, deterministic."""
    import torch.nn as nn
    from Net import kfac
    torch.manual_seed(B + C + S)
    g = torch.randn(B, C, S, S, device="cuda") * 1e-4
    sc = kfac._pow2_scale(g)
    got = kfac._gram_hip(g, nn.Conv2d(8, C, 3, padding=1), 2.0, geometry=(1, 1, 0, 1), in_scale=sc)
    gd = g.double().permute(1, 0, 2, 3).reshape(C, -1)
    want = 2.0 * (gd @ gd.t())
    assert (got.double() - want).abs().max().item() / want.abs().max().item() < 2e-6
    assert torch.equal(got, got.t())
    assert torch.equal(got, kfac._gram_hip(g, nn.Conv2d(8, C, 3, padding=1), 2.0, geometry=(1, 1, 0, 1), in_scale=sc))


@pytest.mark.parametrize("B,C,S", [(160, 64, 4), (160, 64, 6), (160, 32, 2), (2048, 64, 2), (3000, 64, 4)])
def test_gradient_factor_workspace_holds_the_nchw_partials(B, C, S):
    """conv7's gradient factor on 12x12 / 20x20 boards has 16 / 36 positions per image: a Gram plan over B * per rows is smaller
    than k_gram_nchw's min(B, 2048) C x C blocks of partial sums (the kernel wrote up to 1.9 MB past the buffer at
    the reference's 32 envs x 5 steps).  The size query must cover them: a canary right behind the workspace stays intact, and
    the query is at least the blocks' bytes."""
    from Net import kfac
    from tron import _native as nat
    L = nat.lib()
    torch.manual_seed(B + S)
    g = torch.randn(B, C, S, S, device="cuda") * 1e-3
    sc = kfac._pow2_scale(g)
    need = int(L.tron_kfac_patch_gram_workspace(B, C, S, S, 1, 1, 0, 1))
    assert need >= min(B, 2048) * C * C * 4
    tail = 8 << 20
    buf = torch.full((need + tail,), 0x5A, dtype=torch.uint8, device="cuda")
    gram = torch.empty(C, C, device="cuda")
    nat.check(L.tron_kfac_patch_gram(nat.ptr(g), B, C, S, S, 1, 1, 0, 1, 1.0, nat.ptr(sc), nat.ptr(gram), nat.ptr(buf), nat.stream_ptr()),
              "tron_kfac_patch_gram")
    torch.cuda.synchronize()
    assert bool((buf[need:] == 0x5A).all()), "k_gram_nchw wrote past the workspace the size query reserved"
    gd = g.double().permute(1, 0, 2, 3).reshape(C, -1)
    want = gd @ gd.t()
    assert (gram.double() - want).abs().max().item() / want.abs().max().item() < 2e-6


GRAM_SHAPES = [  # B, C, H, W, k, pad, stride
    (5, 3, 12, 12, 3, 1, 1),      # conv1 at 10x10 (d = 27: one ragged tile)
    (9, 4, 34, 34, 3, 1, 1),      # conv1 of MapNet at 32x32
    (40, 32, 12, 12, 3, 1, 1),    # d = 288
    (3, 32, 26, 26, 3, 1, 1),
    (6, 64, 34, 34, 3, 1, 1),     # d = 576, 6 936 rows
    (7, 64, 17, 17, 7, 3, 2),     # conv7 after the pool at 32x32: d = 3 136
    (64, 64, 6, 6, 7, 3, 2),      # conv7 at 10x10: 9 rows per image
    (7, 5, 9, 11, 3, 0, 1),       # no padding, odd sizes, non-square
    (1, 1, 3, 3, 3, 1, 1),
]


@pytest.mark.parametrize("B,C,H,W,k,pad,stride", GRAM_SHAPES)
def test_patch_gram_matches_float64(B, C, H, W, k, pad, stride):
    """tron_kfac_patch_gram == P^T P of F.unfold's patch matrix in float64 (relative to the largest entry), symmetric
    bit for bit (the lower triangle is the mirrored upper one), the same bits when run again."""
    import torch.nn as nn
    import torch.nn.functional as F
    from Net import kfac
    torch.manual_seed(B * 100 + C)
    x = torch.randn(B, C, H, W, device="cuda") * 2.0
    conv = nn.Conv2d(C, 8, k, padding=pad, stride=stride)
    got = kfac._gram_hip(x, conv, 0.5)
    cols = F.unfold(x.double(), (k, k), padding=pad, stride=stride)
    P = cols.transpose(1, 2).reshape(-1, cols.size(1))
    want = 0.5 * (P.t() @ P)
    assert got.shape == want.shape
    assert (got.double() - want).abs().max().item() / want.abs().max().item() < 2e-6
    assert torch.equal(got, got.t())
    assert torch.equal(got, kfac._gram_hip(x, conv, 0.5))


@pytest.mark.parametrize("rows,d", [(512, 32), (1000, 129), (4096, 256), (8192, 576), (3000, 5184), (700, 257)])
def test_linear_gram_matches_float64(rows, d):
    """tron_kfac_gram (a Linear layer's input factor, kfac.py:57-58) == a^T a in float64."""
    import torch.nn as nn
    from Net import kfac
    torch.manual_seed(rows + d)
    a = torch.randn(rows, d, device="cuda") * 3.0
    got = kfac._gram_hip(a, nn.Linear(d, 4), 1.0 / rows)
    want = a.double().t() @ a.double() / rows
    assert (got.double() - want).abs().max().item() / want.abs().max().item() < 2e-6
    assert torch.equal(got, got.t())


def test_gram_over_several_passes_and_bad_arguments():
    """A patch matrix larger than one 512 MB pass (here 64-channel 34x34 inputs: 1.3 GB split) goes through the kernel in
    passes whose partial sums add up: the factor of a batch == the sum of the factors of its halves, each a single pass."""
    import torch.nn as nn
    from Net import kfac
    from tron import _native as nat
    torch.manual_seed(11)
    conv = nn.Conv2d(64, 64, 3, padding=1)
    x = torch.randn(500, 64, 34, 34, device="cuda")
    whole = kfac._gram_hip(x, conv, 1.0)
    halves = kfac._gram_hip(x[:200], conv, 1.0) + kfac._gram_hip(x[200:], conv, 1.0)
    assert (whole - halves).abs().max().item() / whole.abs().max().item() < 1e-6
    L = nat.lib()
    assert L.tron_kfac_gram_workspace(100, 9000) == 0 and L.tron_kfac_patch_gram_workspace(0, 3, 12, 12, 3, 3, 1, 1) == 0
    g = torch.empty(27, 27, device="cuda")
    ws = torch.empty(int(L.tron_kfac_patch_gram_workspace(4, 3, 12, 12, 3, 3, 1, 1)), dtype=torch.uint8, device="cuda")
    xs = torch.randn(4, 3, 12, 12, device="cuda")
    assert L.tron_kfac_patch_gram(None, 4, 3, 12, 12, 3, 3, 1, 1, 1.0, None, nat.ptr(g), nat.ptr(ws), None) == nat.ERR_BAD_ARG
    assert L.tron_kfac_patch_gram(nat.ptr(xs), 4, 3, 12, 12, 3, 3, 1, 0, 1.0, None, nat.ptr(g), nat.ptr(ws), None) == nat.ERR_BAD_ARG
    assert L.tron_kfac_patch_gram(nat.ptr(xs), 0, 3, 12, 12, 3, 3, 1, 1, 1.0, None, nat.ptr(g), nat.ptr(ws), None) == 0
    torch.cuda.synchronize()
    assert torch.count_nonzero(g) == 0                                   # the sum over nothing


def test_cov_inputs_uses_the_gram_kernels_and_matches_the_library_path(monkeypatch):
    """Net/kfac.py::cov_inputs with and without csrc/tron_kfac.hip's Gram kernels, whole batch and micro-batches."""
    import torch.nn as nn
    from Net import kfac
    torch.manual_seed(2)
    conv = nn.Conv2d(32, 8, 3, padding=1)
    lin = nn.Linear(300, 7)
    a, b = torch.randn(600, 32, 12, 12, device="cuda"), torch.randn(2048, 300, device="cuda")
    got_a, got_b = kfac.cov_inputs(a, conv), kfac.cov_inputs(b, lin)
    parts = sum(kfac.cov_inputs(a[i:i + 200], conv, 600) for i in range(0, 600, 200))
    monkeypatch.setattr(kfac, "use_gram", False)
    want_a, want_b = kfac.cov_inputs(a, conv), kfac.cov_inputs(b, lin)
    for got, want in ((got_a, want_a), (got_b, want_b), (parts, want_a)):
        assert (got - want).abs().max().item() / want.abs().max().item() < 1e-5


@pytest.mark.parametrize("magnitude", [1.0, 1e-6, 1e-12])
def test_cov_grads_on_the_gram_kernels_matches_float64(magnitude):
    """kfac.py:61-76 `compute_cov_g` for a convolution's and a Linear layer's output gradients — tensors whose entries sit
    far below f16's range: the kernels scale them by a power of two found on the device (no read-back)."""
    import torch.nn as nn
    from Net import kfac
    torch.manual_seed(4)
    conv, lin = nn.Conv2d(8, 64, 3, padding=1), nn.Linear(10, 128)
    g = torch.randn(300, 64, 12, 12, device="cuda") * magnitude
    h = torch.randn(4096, 128, device="cuda") * magnitude
    for grad, module, whole in ((g, conv, None), (g[:100], conv, 300), (h, lin, None), (h[:1024], lin, 4096)):
        got = kfac.cov_grads(grad, module, whole)
        gd = grad.double()
        batch = grad.size(0) if whole is None else whole
        scale = 1.0 if whole is None else whole / grad.size(0)
        if gd.dim() == 4:
            gd = gd.permute(0, 2, 3, 1).reshape(-1, gd.size(1)) * (gd.size(2) * gd.size(3))
        g_ = gd * batch
        want = g_.t() @ (g_ / (gd.size(0) * scale))
        assert (got.double() - want).abs().max().item() / want.abs().max().item() < 2e-6
    z = kfac.cov_grads(torch.zeros(600, 64, 12, 12, device="cuda"), conv)
    assert torch.count_nonzero(z) == 0 and torch.isfinite(z).all()


@pytest.mark.parametrize("width,model", [(10, "MapNet"), (24, "Mulnet"), (32, "TestNet")])
def test_fused_bias_residual_mish_keeps_kfac_statistics(width, model, monkeypatch):
    """After KFACOptimizer split the biases, a trunk layer is the hooked conv module followed by ONE bias + residual +
    activation pass (Net/kfac.py::SplitBias / AddBias.fused_mish) that feeds AddBias's two statistics hooks by hand: outputs,
    parameter gradients and every Kronecker factor equal those of the module-by-module graph (kfac.py:156-189)."""
    import copy
    import torch.nn.functional as F
    from Net import ACNet, activations, kfac
    torch.manual_seed(width)
    S, B = width + 2, 24
    net = getattr(ACNet, model)(width).cuda()
    net.dropout.p = 0.0
    ref = copy.deepcopy(net)
    opts = [kfac.KFACOptimizer(m) for m in (net, ref)]
    x = torch.randn(B, 4 if model == "MapNet" else 3, S, S, device="cuda")
    extra = () if model == "MapNet" else ((torch.rand(B, 2, device="cuda"),) if model == "Mulnet" else (torch.rand(B, device="cuda"),))
    acts = torch.randint(0, 4, (B, 1), device="cuda")
    outs = []
    for m, opt, fused in ((net, opts[0], True), (ref, opts[1], False)):
        if not fused:
            monkeypatch.setattr(activations, "bias_mish_supported", lambda y, residual=None: False)
        v, logp, ent = m.evaluate_actions(x, acts, *extra)
        opt.acc_stats = True
        (-(logp.mean()) - v.pow(2).mean()).backward(retain_graph=True)
        opt.acc_stats = False
        (v.pow(2).mean() - logp.mean() - 0.01 * ent).backward()
        outs.append((v.detach(), logp.detach()))
    assert (outs[0][0] - outs[1][0]).abs().max().item() < 1e-5 and (outs[0][1] - outs[1][1]).abs().max().item() < 1e-5
    for (name, p), (_, r) in zip(net.named_parameters(), ref.named_parameters()):
        assert (p.grad - r.grad).abs().max().item() / (r.grad.abs().max().item() + 1e-30) < 1e-4, name
    with torch.no_grad():            # gradient-free forwards put bias / residual / activation into the convolution's epilogue
        v0, a0 = net(x, *extra)
    assert (v0 - outs[0][0]).abs().max().item() < 1e-5 and (torch.log_softmax(a0, 1).gather(1, acts) - outs[0][1]).abs().max().item() < 1e-5
    for ma, mb in zip(opts[0].modules, opts[1].modules):
        for store in ("m_aa", "m_gg"):
            a, b = getattr(opts[0], store)[ma], getattr(opts[1], store)[mb]
            assert (a - b).abs().max().item() / (b.abs().max().item() + 1e-30) < 1e-4, (type(ma).__name__, store)


@pytest.mark.parametrize("width", [10, 24, 32])
@pytest.mark.parametrize("split", [False, True])
def test_actor_critic_trunk_on_the_weight_stationary_chain(width, split, monkeypatch):
    """The actor-critic nets' trunk as ONE node on the weight-stationary kernels (`_ACTrunkPX`; gradient-free: `ac_trunk_infer`)
    against the layer-by-layer graph (ACNet.py:97-111), plain (A2C) and after KFACOptimizer split the biases: outputs within
    1e-5, parameter gradients within 1e-4 of their scale, and the gradient-free forward (the rollouts' acting) within 1e-5."""
    import copy
    from Net import ACNet, activations, kfac
    torch.manual_seed(width + split)
    S, B = width + 2, 19
    net = ACNet.Mulnet(width).cuda()
    net.dropout.p = 0.0
    ref = copy.deepcopy(net)
    if split:
        kfac.KFACOptimizer(net), kfac.KFACOptimizer(ref)
    x = torch.randn(B, 3, S, S, device="cuda")
    extra = torch.rand(B, 2, device="cuda")
    acts = torch.randint(0, 4, (B, 1), device="cuda")
    v, logp, ent = net.evaluate_actions(x, acts, extra)
    (v.pow(2).mean() - logp.mean() - 0.01 * ent).backward()
    with torch.no_grad():
        v0, a0 = net(x, extra)
    monkeypatch.setattr(activations, "ac_trunk_px_supported", lambda x, weights, need_grad=True: False)
    vr, logpr, entr = ref.evaluate_actions(x, acts, extra)
    (vr.pow(2).mean() - logpr.mean() - 0.01 * entr).backward()
    with torch.no_grad():
        v0r, a0r = ref(x, extra)
    assert (v - vr).abs().max().item() < 1e-5 and (logp - logpr).abs().max().item() < 1e-5
    assert (v0 - v0r).abs().max().item() < 1e-5 and (a0 - a0r).abs().max().item() < 1e-5 and (v0 - v).abs().max().item() < 1e-5
    for (name, p), (_, r) in zip(net.named_parameters(), ref.named_parameters()):
        assert (p.grad - r.grad).abs().max().item() / (r.grad.abs().max().item() + 1e-30) < 1e-4, name


def test_mish_kernels_match_the_composed_form():
    """csrc/tron_nn.hip: mish forward / backward against x * tanh(softplus(x)) and its autograd gradient
    evaluated in float64, over the whole input range (incl. the > 20 cut-over and deep negatives)."""
    import torch.nn.functional as F
    from Net.activations import mish
    torch.manual_seed(0)
    x = torch.cat([torch.randn(100003) * 3, torch.linspace(-110, 110, 4001), torch.tensor([0.0, 20.0, 20.000002, -0.0, 88.0, -104.0])])
    xg = x.cuda().requires_grad_(True)
    y = mish(xg)
    gy = torch.randn_like(x).cuda()
    (gx,) = torch.autograd.grad(y, xg, gy)
    x64 = x.double().requires_grad_(True)
    y64 = x64 * torch.tanh(F.softplus(x64))
    (gx64,) = torch.autograd.grad(y64, x64, gy.cpu().double())
    err_y = ((y.cpu().double() - y64.detach()).abs() / y64.detach().abs().clamp_min(1e-30)).max().item()
    err_g = ((gx.cpu().double() - gx64).abs() / (gx64.abs() + 0.1)).max().item()     # the derivative crosses zero near -1.19
    assert err_y < 1e-6 and err_g < 2e-6, (err_y, err_g)          # a few fp32 ulps
    # no-grad, non-contiguous and odd-length inputs
    with torch.no_grad():
        z = torch.randn(7, 5, 3, device="cuda").transpose(0, 2)
        assert torch.allclose(mish(z), z * torch.tanh(F.softplus(z)), rtol=1e-6, atol=1e-7)


def test_dqn_net_on_gpu_matches_cpu():
    """The net with the HIP activation (GPU) against the same weights with F.mish (CPU): Q within 1e-5."""
    from Net.DQNNet import Net
    torch.manual_seed(3)
    net = Net(3, 10).eval()
    x = torch.randn(64, 3, 12, 12)
    with torch.no_grad():
        q_cpu = net(x)
        q_gpu = net.cuda()(x.cuda()).cpu()
    assert torch.allclose(q_cpu, q_gpu, rtol=1e-5, atol=1e-5)


def test_fused_bias_mish_path_matches_plain_forward_and_gradients():
    """Net.forward with the fused bias + residual + mish pass behind each convolution against the plain
    composition of the same ops on the same device: outputs and every parameter gradient."""
    from Net.DQNNet import Net
    torch.manual_seed(5)
    net = Net(3, 10).cuda()
    net.dropout.p = 0.0
    x = torch.randn(48, 3, 12, 12, device="cuda")
    w = torch.randn(48, 4, device="cuda")
    (net(x) * w).sum().backward()
    fused = [p.grad.clone() for p in net.parameters()]
    q_fused = net(x).detach()
    net.zero_grad()
    (net._forward_plain(x) * w).sum().backward()
    q_plain = net._forward_plain(x).detach()
    assert torch.allclose(q_fused, q_plain, rtol=1e-5, atol=1e-6)
    for (name, p), g in zip(net.named_parameters(), fused):
        assert torch.allclose(p.grad, g, rtol=2e-4, atol=2e-5), name
    # a 26x26 board (HW = 676) and the 9-cell conv7 output (HW % 4 != 0: falls back to the unfused ops)
    big = Net(3, 24).cuda().eval()
    xb = torch.randn(8, 3, 26, 26, device="cuda")
    with torch.no_grad():
        assert torch.allclose(big(xb), big._forward_plain(xb), rtol=1e-5, atol=1e-6)
