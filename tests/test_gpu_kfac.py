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
