"""Inference path of the reference's CNN trunk on the hand-written HIP kernels (csrc/tron_conv.hip): the six
3x3 convolutions of Net/DQNNet.py:10-17,33-50 (and the identical stacks of Net/ACNet.py), each ONE launch that
does convolution + bias + residual + mish on the fp32 matrix cores, with conv1 reading the env's int8
observation codes directly (the f32 pop_up planes of util.py:11-37 are built inside the kernel).

Used for every forward that needs no gradient: the epsilon-greedy policy over 2N observations per env step
(DDQN.py:90-110) and the two target-side forwards of a learn step (DDQN.py:129-142).  There is no fallback in
here: unsupported shapes are reported by `supported()` and the callers keep the PyTorch modules for those.
"""
import torch

from tron import _native as nat

MATH = {"f32": nat.CONV_F32, "f16x3": nat.CONV_F16X3}
# arithmetic of the convolutions (include/tron_hip.h): "f16x3" = split-f16 matrix cores (12x12 boards; others fall back
# to the f32 kernel inside the library), "f32" = the exact f32 MFMA.  TRON_CONV_MATH overrides.
import os as _os
default_math = _os.environ.get("TRON_CONV_MATH", "f16x3")
_SIDES = (12, 26)            # boards 10x10 and 24x24 (BASELINE configs 2 / 3); tron_conv3x3_fwd's instantiations


def supported(conv, side):
    return (isinstance(conv, torch.nn.Conv2d) and conv.kernel_size == (3, 3) and conv.stride == (1, 1)
            and conv.padding == (1, 1) and conv.dilation == (1, 1) and conv.groups == 1
            and conv.out_channels in (32, 64) and side in _SIDES and conv.weight.is_cuda
            and conv.weight.dtype == torch.float32)


def conv3x3_raw(x, weight, bias=None, residual=None, act=True, codes=False, plane4=0.0, want_pre=False, math=None):
    """act(conv3x3(x, weight, padding=1) + bias + residual) on tensors: weight f32 [Cout, Cin, 3, 3] as nn.Conv2d keeps
    it, x f32 [B, Cin, S, S] — or, with codes=True, int8 observation codes [B, S, S] standing for Cin pop_up planes.
    Returns out (and the pre-activation when want_pre)."""
    B, S = x.shape[0], x.shape[-1]
    cout, cin = weight.shape[0], weight.shape[1]
    x = x.contiguous()
    if codes:
        if x.dtype != torch.int8:
            raise TypeError("codes=True takes the env's int8 observation codes")
    elif x.dtype != torch.float32 or x.shape[1] != cin:
        raise TypeError(f"expected f32 [B, {cin}, S, S], got {tuple(x.shape)} {x.dtype}")
    out = torch.empty(B, cout, S, S, dtype=torch.float32, device=x.device)
    pre = torch.empty_like(out) if want_pre else None
    res = None if residual is None else residual.contiguous()
    b = None if bias is None else bias.detach()
    w = weight.detach()
    if not w.is_contiguous():
        w = w.contiguous()
    m = MATH[math or default_math]
    ws = None
    if m == nat.CONV_F16X3:        # scratch for the split weights, rewritten by every call
        ws = torch.empty(int(nat.lib().tron_conv3x3_workspace(cin, cout)), dtype=torch.uint8, device=x.device)
    with torch.cuda.device(x.device):
        nat.check(nat.lib().tron_conv3x3_fwd(nat.ptr(x), int(codes), nat.ptr(w), nat.ptr(b), nat.ptr(res),
                                             nat.ptr(out), nat.ptr(pre), B, cin, cout, S, float(plane4), int(act),
                                             m, nat.ptr(ws), nat.stream_ptr()), "tron_conv3x3_fwd")
    return (out, pre) if want_pre else out


def conv3x3(x, conv, residual=None, act=True, codes=False, plane4=0.0, want_pre=False, math=None):
    """conv3x3_raw on an nn.Conv2d module."""
    return conv3x3_raw(x, conv.weight, conv.bias, residual, act, codes, plane4, want_pre, math)


def trunk(net, x, codes=False, plane4=0.0):
    """conv1..conv6 with their two residual links (DQNNet.py:34-50): [B, 64, S, S]."""
    x = conv3x3(x, net.conv1, codes=codes, plane4=plane4)
    idx = x
    x = conv3x3(x, net.conv2)
    x = conv3x3(x, net.conv3, residual=idx)
    x = conv3x3(x, net.conv4)
    idx = x
    x = conv3x3(x, net.conv5)
    return conv3x3(x, net.conv6, residual=idx)
