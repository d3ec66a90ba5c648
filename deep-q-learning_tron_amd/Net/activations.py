"""mish(x) = x * tanh(softplus(x)) — the activation of every net in the reference (Net/ACNet.py:56-57).

On the GPU both directions are one pass of csrc/tron_nn.hip (one exp and one division per element: with
e = exp(x), tanh(softplus(x)) = n / (n + 2), n = e (e + 2)); torch's fused F.mish evaluates exp, log1p
and tanh in turn and was 14 % of the DDQN trainer's GPU time.  CPU tensors (the parity tests against the
reference's fixtures) and non-fp32 tensors take F.mish."""
import torch
import torch.nn.functional as F


class _Mish(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        from tron import _native as nat
        x = x.contiguous()
        y = torch.empty_like(x)
        with torch.cuda.device(x.device):
            nat.check(nat.lib().tron_mish_fwd(nat.ptr(x), nat.ptr(y), x.numel(), nat.stream_ptr()), "tron_mish_fwd")
        ctx.save_for_backward(x)
        return y

    @staticmethod
    def backward(ctx, grad_y):
        from tron import _native as nat
        (x,) = ctx.saved_tensors
        g = grad_y.contiguous()
        gx = torch.empty_like(x)
        with torch.cuda.device(x.device):
            nat.check(nat.lib().tron_mish_bwd(nat.ptr(x), nat.ptr(g), nat.ptr(gx), x.numel(), nat.stream_ptr()),
                      "tron_mish_bwd")
        return gx


def mish(x):
    if x.is_cuda and x.dtype == torch.float32 and x.numel() > 0:
        return _Mish.apply(x)
    return F.mish(x)
