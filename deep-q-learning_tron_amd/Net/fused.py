"""The CNN's gradient-free path on the hand-written HIP kernels: the six 3x3 convolutions of Net/DQNNet.py:10-17,33-50
(and the identical stacks of Net/ACNet.py) — each ONE launch that does convolution + bias + residual + mish on the
matrix cores (csrc/tron_conv_f16.hip: split-f16 arithmetic; csrc/tron_conv.hip: exact f32), conv1 reading the env's
int8 observation codes directly (the f32 pop_up planes of util.py:11-37 are built inside the kernel), layers chained
through the split-f16 image — and the rest of the net (pool, conv7, linear layers, arg-max: csrc/tron_head.hip).  Also the
raw wrappers of the gradient kernels (tron_conv3x3_dgrad / _wgrad) that Net/activations.py's autograd functions call.

Used for every forward that needs no gradient: the epsilon-greedy policy over 2N observations per env step
(DDQN.py:90-110) and the two target-side forwards of a learn step (DDQN.py:129-142).  There is no fallback in
here: unsupported shapes are reported by `supported()` / `head_supported()` and the callers keep the PyTorch modules
for those.
"""
import torch

from tron import _native as nat

MATH = {"f32": nat.CONV_F32, "f16x3": nat.CONV_F16X3}
# arithmetic of the convolutions (include/tron_hip.h): "f16x3" = split-f16 matrix cores (12x12 boards; others fall back
# to the f32 kernel inside the library), "f32" = the exact f32 MFMA.  TRON_CONV_MATH overrides.
import os as _os
default_math = _os.environ.get("TRON_CONV_MATH", "f16x3")
_SIDES = (12, 26)            # boards 10x10 and 24x24 (BASELINE configs 2 / 3); tron_conv3x3_fwd's instantiations in both arithmetics
_SIDES_F16X3 = (34,)         # 32x32 boards (config 5, the ACKTR nets): forward / input gradient of the split-f16 kernel only


def side_supported(side):
    return side in _SIDES or (default_math == "f16x3" and side in _SIDES_F16X3)


def supported(conv, side):
    return (isinstance(conv, torch.nn.Conv2d) and conv.kernel_size == (3, 3) and conv.stride == (1, 1)
            and conv.padding == (1, 1) and conv.dilation == (1, 1) and conv.groups == 1
            and conv.out_channels in (32, 64) and side_supported(side) and conv.weight.is_cuda
            and conv.weight.dtype == torch.float32)


class Split16:
    """A layer's output as the split-f16 operand image the next layer of the split kernel stages directly
    (include/tron_hip.h, TRON_CONV_IN_SPLIT16): per image [16-channel chunk][hi | lo][pixel][16 ci] f16."""

    def __init__(self, batch, channels, side, device):
        self.shape = (batch, channels, side, side)
        self.buf = torch.empty(batch * channels * side * side * 4, dtype=torch.uint8, device=device)


def split_weights(convs):
    """The split-f16 weight images of several layers in ONE launch (tron_conv3x3_split_weights): a list of uint8 workspace
    views, one per layer, to hand to conv3x3(..., presplit=ws).  Built afresh by every forward pass — nothing is cached."""
    import ctypes as C
    L = nat.lib()
    n = len(convs)
    weights = [c if torch.is_tensor(c) else c.weight for c in convs]          # modules or their weight tensors [Cout, Cin, 3, 3]
    dev = weights[0].device
    sizes = [(int(L.tron_conv3x3_workspace(w.shape[1], w.shape[0])) + 255) // 256 * 256 for w in weights]
    buf = torch.empty(sum(sizes), dtype=torch.uint8, device=dev)
    views, off = [], 0
    for sz in sizes:
        views.append(buf[off:off + sz])
        off += sz
    ws = [w.detach() if w.is_contiguous() else w.detach().contiguous() for w in weights]
    wp = (C.c_void_p * n)(*[w.data_ptr() for w in ws])
    vp = (C.c_void_p * n)(*[v.data_ptr() for v in views])
    ci = (C.c_int32 * n)(*[w.shape[1] for w in weights])
    co = (C.c_int32 * n)(*[w.shape[0] for w in weights])
    with torch.cuda.device(dev):
        nat.check(L.tron_conv3x3_split_weights(wp, ci, co, vp, n, nat.stream_ptr()), "tron_conv3x3_split_weights")
    return views


def conv3x3_raw(x, weight, bias=None, residual=None, act=True, codes=False, plane4=0.0, want_pre=False, math=None,
                want_f32=True, want_split=False, presplit=None):
    """act(conv3x3(x, weight, padding=1) + bias + residual) on tensors: weight f32 [Cout, Cin, 3, 3] as nn.Conv2d keeps
    it; x f32 [B, Cin, S, S], or (codes=True) int8 observation codes [B, S, S] standing for Cin pop_up planes, or a
    Split16 from the previous layer.  Returns the f32 output (None if want_f32=False), then — when asked — the
    pre-activation and / or the Split16 image of the output (want_split; split-f16 arithmetic only)."""
    cout, cin = weight.shape[0], weight.shape[1]
    if isinstance(x, Split16):
        B, S = x.shape[0], x.shape[-1]
        if x.shape[1] != cin:
            raise TypeError(f"expected {cin} channels, got {x.shape[1]}")
        in_fmt, xin, dev = nat.CONV_IN_SPLIT16, x.buf, x.buf.device
    else:
        B, S = x.shape[0], x.shape[-1]
        xin, dev = x.contiguous(), x.device
        if codes:
            if xin.dtype != torch.int8 or xin.dim() != 3 or xin.shape[-2] != S:
                raise TypeError(f"codes=True takes the env's int8 observation codes [B, S, S], got {tuple(xin.shape)} {xin.dtype}")
            in_fmt = nat.CONV_IN_CODES
        else:
            if xin.dtype != torch.float32 or xin.dim() != 4 or xin.shape[1] != cin or xin.shape[-2] != S:
                raise TypeError(f"expected f32 [B, {cin}, S, S], got {tuple(xin.shape)} {xin.dtype}")
            in_fmt = nat.CONV_IN_F32
    out = torch.empty(B, cout, S, S, dtype=torch.float32, device=dev) if want_f32 else None
    pre = torch.empty(B, cout, S, S, dtype=torch.float32, device=dev) if want_pre else None
    split = Split16(B, cout, S, dev) if want_split else None
    res = None if residual is None else residual.contiguous()
    b = None if bias is None else bias.detach()
    w = weight.detach()
    if not w.is_contiguous():
        w = w.contiguous()
    m = MATH[math or default_math]
    ws = None
    if m == nat.CONV_F16X3 and presplit is not None:      # this forward pass split its weights already (split_weights)
        m, ws = nat.CONV_F16X3_PRESPLIT, presplit
    elif m == nat.CONV_F16X3:      # scratch for the split weights, rewritten by every call
        ws = torch.empty(int(nat.lib().tron_conv3x3_workspace(cin, cout)), dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        nat.check(nat.lib().tron_conv3x3_fwd(nat.ptr(xin), in_fmt, nat.ptr(w), nat.ptr(b), nat.ptr(res),
                                             nat.ptr(out), nat.ptr(pre), B, cin, cout, S, float(plane4), int(act),
                                             m, nat.ptr(ws), nat.ptr(None if split is None else split.buf),
                                             nat.stream_ptr()), "tron_conv3x3_fwd")
    ret = [out]
    if want_pre:
        ret.append(pre)
    if want_split:
        ret.append(split)
    return ret[0] if len(ret) == 1 else tuple(ret)


def conv3x3(x, conv, residual=None, act=True, codes=False, plane4=0.0, want_pre=False, math=None, want_f32=True,
            want_split=False, presplit=None):
    """conv3x3_raw on an nn.Conv2d module."""
    return conv3x3_raw(x, conv.weight, conv.bias, residual, act, codes, plane4, want_pre, math, want_f32, want_split,
                       presplit)


def trunk(net, x, codes=False, plane4=0.0, math=None):
    """conv1..conv6 with their two residual links (DQNNet.py:34-50): [B, 64, S, S].  With the split-f16 arithmetic the
    layers hand their outputs on as Split16 images (no re-splitting in the consumer); f32 tensors are written only where
    somebody reads them: conv1's and conv4's outputs (the residuals of conv3 and conv6) and conv6's."""
    if MATH[math or default_math] == nat.CONV_F16X3 and codes and ws_supported(net, x.shape[-1]):
        return trunk_px(net, x, plane4)
    if MATH[math or default_math] != nat.CONV_F16X3:
        x = conv3x3(x, net.conv1, codes=codes, plane4=plane4, math=math)
        idx = x
        x = conv3x3(x, net.conv2, math=math)
        x = conv3x3(x, net.conv3, residual=idx, math=math)
        x = conv3x3(x, net.conv4, math=math)
        idx = x
        x = conv3x3(x, net.conv5, math=math)
        return conv3x3(x, net.conv6, residual=idx, math=math)
    w = split_weights([net.conv1, net.conv2, net.conv3, net.conv4, net.conv5, net.conv6])      # one launch for all six
    idx, s = conv3x3(x, net.conv1, codes=codes, plane4=plane4, math=math, want_split=True, presplit=w[0])
    _, s = conv3x3(s, net.conv2, math=math, want_f32=False, want_split=True, presplit=w[1])
    _, s = conv3x3(s, net.conv3, residual=idx, math=math, want_f32=False, want_split=True, presplit=w[2])
    idx, s = conv3x3(s, net.conv4, math=math, want_split=True, presplit=w[3])
    _, s = conv3x3(s, net.conv5, math=math, want_f32=False, want_split=True, presplit=w[4])
    return conv3x3(s, net.conv6, residual=idx, math=math, presplit=w[5])


# ---- the weight-stationary chain (csrc/tron_conv_ws.hip): activations as PX16 images between the layers ---------------
use_ws = _os.environ.get("TRON_CONV_WS", "1") != "0"
_WS_SHAPES = ((32, 32), (32, 64), (64, 64))


class PX16:
    """An activation tensor [batch, channels, side, side] as the pixel-major split-f16 image of include/tron_hip.h:
    per image [hi | lo][channel octet][pixel][8 channels] f16, value / 64 = hi + lo 2^-11."""

    def __init__(self, batch, channels, side, device):
        self.shape = (batch, channels, side, side)
        self.buf = torch.empty(batch * channels * side * side * 4, dtype=torch.uint8, device=device)

    @classmethod
    def from_f32(cls, x):
        """The PX16 image of an f32 NCHW tensor (tron_px16_from_f32)."""
        B, C, S, _ = x.shape
        out = cls(B, C, S, x.device)
        with torch.cuda.device(x.device):
            nat.check(nat.lib().tron_px16_from_f32(nat.ptr(x.contiguous()), nat.ptr(out.buf), B, C, S, nat.stream_ptr()), "tron_px16_from_f32")
        return out

    def kfac_input_gram(self, scale):
        """scale * P^T P of a 3x3 / pad 1 / stride 1 layer whose input this image is (tron_kfac_gram_px16); None: shape not covered."""
        B, C, S, _ = self.shape
        L, dev = nat.lib(), self.buf.device
        nbytes = int(L.tron_kfac_gram_px16_workspace(B, C, S))
        if nbytes <= 0:
            return None
        gram = torch.empty(9 * C, 9 * C, dtype=torch.float32, device=dev)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        with torch.cuda.device(dev):
            nat.check(L.tron_kfac_gram_px16(nat.ptr(self.buf), B, C, S, float(scale), nat.ptr(gram), nat.ptr(ws), nat.stream_ptr()), "tron_kfac_gram_px16")
        return gram

    def float(self):
        """The f32 NCHW tensor (tron_px16_to_f32)."""
        B, C, S, _ = self.shape
        out = torch.empty(self.shape, dtype=torch.float32, device=self.buf.device)
        with torch.cuda.device(self.buf.device):
            nat.check(nat.lib().tron_px16_to_f32(nat.ptr(self.buf), nat.ptr(out), B, C, S, nat.stream_ptr()), "tron_px16_to_f32")
        return out


def ws_supported(net, side):
    """The weight-stationary chain covers the DQN trunk's shapes (DQNNet.py:10-15) on 12x12 and 26x26 observations."""
    convs = [getattr(net, f"conv{i}", None) for i in range(1, 7)]
    return (use_ws and side in (12, 26) and all(isinstance(c, torch.nn.Conv2d) for c in convs)
            and convs[0].in_channels in (3, 4) and convs[0].out_channels == 32
            and all(c.kernel_size == (3, 3) and c.stride == (1, 1) and c.padding == (1, 1) and c.dilation == (1, 1)
                    and c.groups == 1 and c.bias is not None and c.weight.is_cuda and c.weight.dtype == torch.float32
                    and c.weight.is_contiguous() for c in convs)
            and all((c.in_channels, c.out_channels) in _WS_SHAPES for c in convs[1:]))


def ws_split_weights(convs):
    """The MFMA fragment images of several layers' weights in ONE launch (tron_conv3x3_ws_split_weights): a list of uint8
    workspace views to hand to conv_ws(..., wfrag=).  Built afresh by every forward pass — nothing is cached."""
    import ctypes as C
    L = nat.lib()
    n = len(convs)
    dev = convs[0].weight.device
    sizes = [(int(L.tron_conv3x3_ws_workspace(c.in_channels, c.out_channels)) + 255) // 256 * 256 for c in convs]
    buf = torch.empty(sum(sizes), dtype=torch.uint8, device=dev)
    views, off = [], 0
    for sz in sizes:
        views.append(buf[off:off + sz])
        off += sz
    ws = [c.weight.detach() if c.weight.is_contiguous() else c.weight.detach().contiguous() for c in convs]
    wp = (C.c_void_p * n)(*[w.data_ptr() for w in ws])
    vp = (C.c_void_p * n)(*[v.data_ptr() for v in views])
    ci = (C.c_int32 * n)(*[c.in_channels for c in convs])
    co = (C.c_int32 * n)(*[c.out_channels for c in convs])
    with torch.cuda.device(dev):
        nat.check(L.tron_conv3x3_ws_split_weights(wp, ci, co, vp, n, nat.stream_ptr()), "tron_conv3x3_ws_split_weights")
    return views


def conv1_px16(codes, conv, plane4=0.0):
    """mish(conv1(pop_up(codes)) + bias) from the env's int8 observation codes [B, S, S] -> PX16 with 32 channels."""
    if codes.dtype != torch.int8 or codes.dim() != 3 or codes.shape[-1] != codes.shape[-2]:
        raise TypeError("conv1_px16 takes int8 observation codes [B, S, S]")
    c = codes.contiguous()
    B, S = c.shape[0], c.shape[-1]
    out = PX16(B, conv.out_channels, S, c.device)
    w = conv.weight.detach()
    with torch.cuda.device(c.device):
        nat.check(nat.lib().tron_conv1_px16(nat.ptr(c), nat.ptr(w if w.is_contiguous() else w.contiguous()),
                                            nat.ptr(conv.bias.detach()), conv.in_channels, float(plane4), B, S,
                                            nat.ptr(out.buf), nat.stream_ptr()), "tron_conv1_px16")
    return out


def conv_ws(x, conv, wfrag, residual=None, act=True, want_px=True, want_f32=False, want_pre=False):
    """act(conv3x3(x) + bias + residual) on PX16 images (tron_conv3x3_ws_fwd).  Returns the PX16 output and / or the f32
    NCHW output / pre-activation, in that order, for what was asked."""
    B, cin, S, _ = x.shape
    cout = conv.out_channels
    if cin != conv.in_channels or (residual is not None and residual.shape != (B, cout, S, S)):
        raise TypeError("conv_ws: channel / shape mismatch")
    dev = x.buf.device
    out = PX16(B, cout, S, dev) if want_px else None
    o32 = torch.empty(B, cout, S, S, dtype=torch.float32, device=dev) if want_f32 else None
    pre = torch.empty(B, cout, S, S, dtype=torch.float32, device=dev) if want_pre else None
    with torch.cuda.device(dev):
        nat.check(nat.lib().tron_conv3x3_ws_fwd(nat.ptr(x.buf), nat.ptr(wfrag), nat.ptr(conv.bias.detach()),
                                                nat.ptr(None if residual is None else residual.buf),
                                                nat.ptr(None if out is None else out.buf), nat.ptr(o32), nat.ptr(pre),
                                                B, cin, cout, S, int(act), nat.stream_ptr()), "tron_conv3x3_ws_fwd")
    ret = [t for t, want in ((out, want_px), (o32, want_f32), (pre, want_pre)) if want]
    return ret[0] if len(ret) == 1 else tuple(ret)


use_pool_fused = _os.environ.get("TRON_POOL_FUSED", "1") != "0"      # 12x12: conv6 and the pooling as one launch (0: two) — gradient-free forwards
use_pool_fused_train = use_pool_fused                                  # ... and the learner's forward


class Pooled12:
    """conv6's output of a 12x12 observation batch after AvgPool2d(3, 2, 1) (DQNNet.py:52), as the split-f16 rows conv7's GEMM
    reads: what tron_conv3x3_ws_fwd_pool12 writes and tron_dqn_head_fwd_pooled takes."""

    def __init__(self, batch, device):
        self.shape = (batch, 64, 6, 6)
        self.buf = torch.empty(int(nat.lib().tron_pooled12_bytes(batch)), dtype=torch.uint8, device=device)


def conv_ws_pool12(x, conv, wfrag, residual):
    """AvgPool2d(3, 2, 1)(mish(conv3x3(x) + bias + residual)) of 64-channel 12x12 PX16 images in one launch (tron_conv3x3_ws_fwd_pool12):
    the convolution's output stays in LDS.  Returns a Pooled12."""
    B, cin, S, _ = x.shape
    if (cin, conv.in_channels, conv.out_channels, S) != (64, 64, 64, 12) or residual is None or residual.shape != x.shape:
        raise TypeError("conv_ws_pool12: 64 -> 64 channels at 12x12 with a residual")
    out = Pooled12(B, x.buf.device)
    with torch.cuda.device(x.buf.device):
        nat.check(nat.lib().tron_conv3x3_ws_fwd_pool12(nat.ptr(x.buf), nat.ptr(wfrag), nat.ptr(conv.bias.detach()), nat.ptr(residual.buf),
                                                       nat.ptr(out.buf), B, nat.stream_ptr()), "tron_conv3x3_ws_fwd_pool12")
    return out


def conv_ws_infer(x, cout, wfrag, bias, residual=None, want_f32=False):
    """mish(conv3x3(x) + bias + residual), gradient-free, from tensors instead of a module: x PX16 -> PX16, or (want_f32) the f32
    NCHW tensor (tron_conv3x3_ws_fwd)."""
    B, cin, S, _ = x.shape
    dev = x.buf.device
    out = None if want_f32 else PX16(B, cout, S, dev)
    o32 = torch.empty(B, cout, S, S, dtype=torch.float32, device=dev) if want_f32 else None
    with torch.cuda.device(dev):
        nat.check(nat.lib().tron_conv3x3_ws_fwd(nat.ptr(x.buf), nat.ptr(wfrag), nat.ptr(bias.detach()), nat.ptr(None if residual is None else residual.buf),
                                                nat.ptr(None if out is None else out.buf), nat.ptr(o32), None, B, cin, cout, S, 1, nat.stream_ptr()),
                  "tron_conv3x3_ws_fwd")
    return o32 if want_f32 else out


def trunk_px(net, codes, plane4=0.0, want="f32"):
    """conv1..conv6 with their two residual links (DQNNet.py:34-50) from the env's int8 codes [B, S, S], the
    activations staying PX16 images from conv1's output to conv6's.  want: "f32" -> [B, 64, S, S], "px16" -> PX16, "head" -> what
    `head` takes with the least traffic: at 12x12 the pooled rows (conv6 and the pooling in one launch), else the PX16 image."""
    w = ws_split_weights([net.conv2, net.conv3, net.conv4, net.conv5, net.conv6])          # one launch for all five
    a = conv1_px16(codes, net.conv1, plane4)
    b = conv_ws(a, net.conv2, w[0])
    c = conv_ws(b, net.conv3, w[1], residual=a)
    del a, b                                      # (an image is 4 bytes per element: 23 GB for 131 072 x 64 x 26 x 26 — let the allocator reuse them)
    d = conv_ws(c, net.conv4, w[2])
    del c
    e = conv_ws(d, net.conv5, w[3])
    if want == "head" and use_pool_fused and e.shape[-1] == 12 and e.shape[0] > 0:
        return conv_ws_pool12(e, net.conv6, w[4], d)
    if want in ("px16", "head"):
        return conv_ws(e, net.conv6, w[4], residual=d)
    return conv_ws(e, net.conv6, w[4], residual=d, want_px=False, want_f32=True)


# ---- the learner's trunk on the weight-stationary design (csrc/tron_conv_ws_train.hip): forward keeps a_k and z_k as PX16 images,
# ---- gradients travel as PX16 "gradient images" (value * a power-of-two scale kept in a device record) -------------------
use_trunk_px = _os.environ.get("TRON_TRUNK_PX", "1") != "0"


class GradPX:
    """A gradient tensor [batch, channels, side, side] as a PX16 image of g * s, with its device record info = {s, 1 / s, -, -,
    max |g| per channel [64]} (include/tron_hip.h): s is a power of two chosen before the producing kernel runs."""

    def __init__(self, batch, channels, side, device):
        self.shape = (batch, channels, side, side)
        self.buf = torch.empty(batch * channels * side * side * 4, dtype=torch.uint8, device=device)
        self.info = torch.empty(68, dtype=torch.float32, device=device)          # (written by the producing kernels: {s, 1 / s} and the channels' maxima)

    def float(self):
        px = PX16.__new__(PX16)
        px.shape, px.buf = self.shape, self.buf
        return px.float() * self.info[1]


def _px(shape, buf):
    px = PX16.__new__(PX16)
    px.shape, px.buf = tuple(shape), buf
    return px


def _split_jobs(weights, fn_name, transposed):
    """Fragment images of several weight tensors [cout, cin, 3, 3] in ONE launch; transposed: the input gradient's
    (rotated, transposed) fragments, plus the bound factors wnorm [n]."""
    import ctypes as C
    L = nat.lib()
    n = len(weights)
    dev = weights[0].device
    shapes = [(w.shape[1], w.shape[0]) for w in weights]                                  # (cin, cout) of the forward layers
    sizes = [(int(L.tron_conv3x3_ws_workspace(*((co, ci) if transposed else (ci, co)))) + 255) // 256 * 256 for ci, co in shapes]
    buf = torch.empty(sum(sizes), dtype=torch.uint8, device=dev)
    views, off = [], 0
    for sz in sizes:
        views.append(buf[off:off + sz])
        off += sz
    ws = [w.detach() if w.is_contiguous() else w.detach().contiguous() for w in weights]
    wp = (C.c_void_p * n)(*[w.data_ptr() for w in ws])
    vp = (C.c_void_p * n)(*[v.data_ptr() for v in views])
    ci = (C.c_int32 * n)(*[c for c, _ in shapes])
    co = (C.c_int32 * n)(*[c for _, c in shapes])
    with torch.cuda.device(dev):
        if transposed:
            wn = torch.empty(n, dtype=torch.float32, device=dev)
            nat.check(L.tron_conv3x3_ws_split_weights_bwd(wp, ci, co, vp, nat.ptr(wn), n, nat.stream_ptr()), fn_name)
            return views, wn
        nat.check(L.tron_conv3x3_ws_split_weights(wp, ci, co, vp, n, nat.stream_ptr()), fn_name)
    return views


def conv1_px16_train(codes, weight, bias, plane4=0.0):
    """mish(conv1(pop_up(codes)) + bias) and its pre-activation, both PX16 [B, 32, S, S] (tron_conv1_px16_train)."""
    c = codes.contiguous()
    B, S = c.shape[0], c.shape[-1]
    a, z = PX16(B, weight.shape[0], S, c.device), PX16(B, weight.shape[0], S, c.device)
    w = weight.detach()
    with torch.cuda.device(c.device):
        nat.check(nat.lib().tron_conv1_px16_train(nat.ptr(c), nat.ptr(w if w.is_contiguous() else w.contiguous()), nat.ptr(bias.detach()),
                                                  weight.shape[1], float(plane4), B, S, nat.ptr(a.buf), nat.ptr(z.buf), nat.stream_ptr()),
                  "tron_conv1_px16_train")
    return a, z


def conv_ws_train(x, cout, wfrag, bias, residual=None, want_f32=False):
    """mish(conv3x3(x) + bias + residual) with the pre-activation kept: x PX16 -> (PX16 output — or, want_f32, the f32 NCHW
    output the head reads —, PX16 pre-activation)   (tron_conv3x3_ws_train_fwd)."""
    B, cin, S, _ = x.shape
    dev = x.buf.device
    z = PX16(B, cout, S, dev)
    out = None if want_f32 else PX16(B, cout, S, dev)
    o32 = torch.empty(B, cout, S, S, dtype=torch.float32, device=dev) if want_f32 else None
    with torch.cuda.device(dev):
        nat.check(nat.lib().tron_conv3x3_ws_train_fwd(nat.ptr(x.buf), nat.ptr(wfrag), nat.ptr(bias.detach()),
                                                      nat.ptr(None if residual is None else residual.buf), nat.ptr(None if out is None else out.buf),
                                                      nat.ptr(o32), nat.ptr(z.buf), B, cin, cout, S, nat.stream_ptr()), "tron_conv3x3_ws_train_fwd")
    return (o32 if want_f32 else out), z


def conv_ws_train_pool12(x, wfrag, bias, residual):
    """The learner's conv6 + pooling at 12x12 in one launch (tron_conv3x3_ws_train_fwd_pool12): x, residual PX16 [B, 64, 12, 12] ->
    (AvgPool2d(3, 2, 1)(mish(conv3x3(x) + bias + residual)) as f32 [B, 64 * 36], PX16 pre-activation).  The activation itself is
    not stored: in the DQN net nothing but the pooling reads it (DQNNet.py:48-52)."""
    B, cin, S, _ = x.shape
    if (cin, S) != (64, 12) or residual is None or residual.shape != x.shape:
        raise TypeError("conv_ws_train_pool12: 64 -> 64 channels at 12x12 with a residual")
    dev = x.buf.device
    z = PX16(B, 64, 12, dev)
    pooled = torch.empty(B, 64 * 36, dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        nat.check(nat.lib().tron_conv3x3_ws_train_fwd_pool12(nat.ptr(x.buf), nat.ptr(wfrag), nat.ptr(bias.detach()), nat.ptr(residual.buf),
                                                             nat.ptr(z.buf), nat.ptr(pooled), B, nat.stream_ptr()), "tron_conv3x3_ws_train_fwd_pool12")
    return pooled, z


def grad_px_from_f32(g, z):
    """The gradient chain's entry: (g * mish'(z)) as a gradient image + the column sums (the layer's bias gradient);
    g f32 [B, C, S, S], z the layer's PX16 pre-activation."""
    L = nat.lib()
    B, C, S, _ = g.shape
    dev = g.device
    sc = torch.zeros(4, dtype=torch.float32, device=dev)
    out = GradPX(B, C, S, dev)
    gb = torch.empty(C, dtype=torch.float32, device=dev)
    ws = torch.empty(int(L.tron_px16_grad_workspace(B, C)), dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        nat.check(L.tron_absmax_pow2(nat.ptr(g), g.numel(), 14, nat.ptr(sc), nat.stream_ptr()), "tron_absmax_pow2")
        nat.check(L.tron_px16_grad_from_f32(nat.ptr(g), nat.ptr(z.buf), nat.ptr(sc), B, C, S, nat.ptr(out.buf), nat.ptr(out.info), nat.ptr(gb),
                                            nat.ptr(ws), nat.stream_ptr()), "tron_px16_grad_from_f32")
    return out, gb


def conv_ws_dgrad(gp, cin, wfrag_rot, wnorm, z_below, extra=None, want_px=True, want_f32=False):
    """(conv^T(gp, W) + extra) * mish'(z_below): gp a GradPX with the forward layer's cout channels -> (GradPX with cin channels
    or None, f32 NCHW or None, bias gradient of the layer below [cin])   (tron_conv3x3_ws_dgrad)."""
    L = nat.lib()
    B, cout, S, _ = gp.shape
    dev = gp.buf.device
    out = GradPX(B, cin, S, dev)
    if not want_px:
        out.buf = None
    o32 = torch.empty(B, cin, S, S, dtype=torch.float32, device=dev) if want_f32 else None
    gb = torch.empty(cin, dtype=torch.float32, device=dev)
    nbytes = int(L.tron_conv3x3_ws_dgrad_workspace(cin, cout))
    if nbytes <= 0:
        raise nat.TronNativeError(f"tron_conv3x3_ws_dgrad: no kernel for cin={cin} cout={cout}")
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        nat.check(L.tron_conv3x3_ws_dgrad(nat.ptr(gp.buf), nat.ptr(gp.info), nat.ptr(wfrag_rot), nat.ptr(wnorm),
                                          nat.ptr(None if extra is None else extra.buf), nat.ptr(None if extra is None else extra.info),
                                          nat.ptr(z_below.buf), nat.ptr(out.buf), nat.ptr(o32), nat.ptr(out.info), nat.ptr(gb), B, cin, cout, S,
                                          nat.ptr(ws), nat.stream_ptr()), "tron_conv3x3_ws_dgrad")
    return (out if want_px else None), o32, gb


def wgrad_px_supported(cin, cout, side):
    return side in (12, 26, 34) and (cin, cout) in _WS_SHAPES


def conv3x3_wgrad_px(a, gp):
    """Weight gradient of conv3x3 from the layer's input a (PX16) and the gradient image at its output -> f32 [cout, cin, 3, 3]
    (tron_conv3x3_wgrad_px16)."""
    L = nat.lib()
    B, cin, S, _ = a.shape
    cout = gp.shape[1]
    dev = a.buf.device
    nbytes = int(L.tron_conv3x3_wgrad_px16_workspace(B, cin, cout, S))
    if nbytes <= 0:
        raise nat.TronNativeError(f"tron_conv3x3_wgrad_px16: no kernel for cin={cin} cout={cout} side={S}")
    gw = torch.empty(cout, cin, 3, 3, dtype=torch.float32, device=dev)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        nat.check(L.tron_conv3x3_wgrad_px16(nat.ptr(a.buf), nat.ptr(gp.buf), nat.ptr(gp.info), nat.ptr(gw), B, cin, cout, S, nat.ptr(ws),
                                            nat.stream_ptr()), "tron_conv3x3_wgrad_px16")
    return gw


def conv1_wgrad_px(codes, gp, cin, plane4=0.0):
    """conv1's weight gradient from the int8 codes [B, S, S] and the gradient image at its pre-activation (32 channels) -> f32
    [32, cin, 3, 3] (tron_conv1_wgrad_px16): no f32 planes, no f32 gradient."""
    L = nat.lib()
    B, S = codes.shape[0], codes.shape[-1]
    dev = gp.buf.device
    nbytes = int(L.tron_conv1_wgrad_px16_workspace(B, S))
    if nbytes <= 0 or gp.shape[1] != 32 or cin not in (3, 4):
        raise nat.TronNativeError(f"tron_conv1_wgrad_px16: no kernel for side={S} cin={cin}")
    gw = torch.empty(32, cin, 3, 3, dtype=torch.float32, device=dev)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        nat.check(L.tron_conv1_wgrad_px16(nat.ptr(codes), nat.ptr(gp.buf), nat.ptr(gp.info), B, S, cin, float(plane4), nat.ptr(gw), nat.ptr(ws),
                                          nat.stream_ptr()), "tron_conv1_wgrad_px16")
    return gw


def conv3x3_dgrad(gp, weight, absmax=None):
    """Input gradient of conv3x3(x, weight, padding=1): gp f32 [B, Cout, S, S] (the gradient at the convolution's output),
    weight the FORWARD layer's [Cout, Cin, 3, 3] -> f32 [B, Cin, S, S] (tron_conv3x3_dgrad).  absmax: per-block maxima of
    |gp| from bias_mish_bwd — the gradient is scaled by the power of two they give on its way into f16."""
    L = nat.lib()
    B, cout, side, _ = gp.shape
    cin = weight.shape[1]
    assert gp.is_contiguous() and gp.dtype == torch.float32 and tuple(weight.shape) == (cout, cin, 3, 3)
    w = weight if weight.is_contiguous() else weight.contiguous()
    gx = torch.empty(B, cin, side, side, dtype=torch.float32, device=gp.device)
    ws = torch.empty(max(16, int(L.tron_conv3x3_workspace(cout, cin))), dtype=torch.uint8, device=gp.device)
    with torch.cuda.device(gp.device):
        nat.check(L.tron_conv3x3_dgrad(gp.data_ptr(), w.data_ptr(), None if absmax is None else absmax.data_ptr(),
                                       0 if absmax is None else absmax.numel(), gx.data_ptr(), B, cin, cout, side,
                                       ws.data_ptr(), torch.cuda.current_stream(gp.device).cuda_stream), "tron_conv3x3_dgrad")
    return gx


def dgrad_mish_supported(weight, side):
    """tron_conv3x3_dgrad_mish's shapes (conv2..conv6 of the DQN trunk at 12x12 and 26x26)."""
    return (side in _SIDES and tuple(weight.shape[2:]) == (3, 3) and weight.shape[0] in (32, 64) and weight.shape[1] in (32, 64)
            and weight.is_cuda and weight.dtype == torch.float32)


def conv3x3_dgrad_mish(gp, weight, absmax, pre_below, extra=None):
    """(dgrad(gp, weight) + extra) * mish'(pre_below) in one launch, with the bias gradient of the layer below and the
    per-channel maxima of the result (tron_conv3x3_dgrad_mish): gp f32 [B, Cout, S, S], weight the FORWARD layer's
    [Cout, Cin, 3, 3], pre_below / extra f32 [B, Cin, S, S] -> (grad_pre_below, bias_grad_below [Cin], absmax_below [Cin])."""
    L = nat.lib()
    B, cout, side, _ = gp.shape
    cin = weight.shape[1]
    assert gp.is_contiguous() and gp.dtype == torch.float32 and tuple(weight.shape) == (cout, cin, 3, 3)
    assert pre_below.is_contiguous() and tuple(pre_below.shape) == (B, cin, side, side) and pre_below.dtype == torch.float32
    assert extra is None or (extra.is_contiguous() and extra.shape == pre_below.shape and extra.dtype == torch.float32)
    w = weight if weight.is_contiguous() else weight.contiguous()
    dev = gp.device
    out = torch.empty_like(pre_below)
    gb = torch.empty(cin, dtype=torch.float32, device=dev)
    am = torch.empty(cin, dtype=torch.float32, device=dev)
    nbytes = int(L.tron_conv3x3_dgrad_mish_workspace(B, cin, cout, side))
    if nbytes <= 0:
        raise nat.TronNativeError(f"tron_conv3x3_dgrad_mish: no kernel for cin={cin} cout={cout} side={side}")
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        nat.check(L.tron_conv3x3_dgrad_mish(gp.data_ptr(), w.data_ptr(), None if absmax is None else absmax.data_ptr(),
                                            0 if absmax is None else absmax.numel(), None if extra is None else extra.data_ptr(),
                                            pre_below.data_ptr(), out.data_ptr(), gb.data_ptr(), am.data_ptr(), B, cin, cout, side,
                                            ws.data_ptr(), torch.cuda.current_stream(dev).cuda_stream), "tron_conv3x3_dgrad_mish")
    return out, gb, am


def wgrad_supported(weight, side):
    """tron_conv3x3_wgrad's shapes: every layer of the trunk at 12x12; conv2..conv6 at 26x26 and 34x34 (24x24 / 32x32 boards: the
    row-streaming kernel, a 34-pixel row as two column halves); conv1 (3 or 4 planes -> 32) there as plain f32 FMAs."""
    co, ci = weight.shape[0], weight.shape[1]
    shape_ok = ((side == 12 and ci in (3, 4, 32, 64)) or (side in (26, 34) and (ci, co) in ((32, 32), (32, 64), (64, 64)))
                or (side in (26, 34) and ci in (3, 4) and co == 32))
    return (shape_ok and co in (32, 64) and tuple(weight.shape[2:]) == (3, 3) and weight.is_cuda
            and weight.dtype == torch.float32)


def conv3x3_wgrad(x, gp, absmax=None):
    """Weight gradient of conv3x3(x, W, padding=1): x f32 [B, Cin, S, S], gp f32 [B, Cout, S, S] -> f32 [Cout, Cin, 3, 3]
    (tron_conv3x3_wgrad; absmax as in conv3x3_dgrad, None = the library finds the maximum itself)."""
    L = nat.lib()
    B, cin, side, _ = x.shape
    cout = gp.shape[1]
    assert x.is_contiguous() and gp.is_contiguous() and x.dtype == gp.dtype == torch.float32 and gp.shape[0] == B
    gw = torch.empty(cout, cin, 3, 3, dtype=torch.float32, device=x.device)
    ws = torch.empty(int(L.tron_conv3x3_wgrad_workspace(cin, cout)), dtype=torch.uint8, device=x.device)
    with torch.cuda.device(x.device):
        nat.check(L.tron_conv3x3_wgrad(x.data_ptr(), gp.data_ptr(), None if absmax is None else absmax.data_ptr(),
                                       0 if absmax is None else absmax.numel(), gw.data_ptr(), B, cin, cout, side,
                                       ws.data_ptr(), torch.cuda.current_stream(x.device).cuda_stream), "tron_conv3x3_wgrad")
    return gw


def head_supported(net, side):
    """tron_dqn_head_fwd covers the reference's own geometry — 12x12 observations, 64*3*3 into fc1 (DQNNet.py:24,55) — and
    the 24x24 boards' (26x26 observations, 64*7*7)."""
    flat = {12: 576, 26: 3136}.get(side)              # 64 x 3 x 3 (10x10 boards, the reference's own) / 64 x 7 x 7 (24x24)
    return (flat is not None and getattr(net, "flat", 0) == flat and net.conv7.weight.shape == (64, 64, 7, 7)
            and net.fc1.weight.shape == (256, flat) and net.fc2.weight.shape == (128, 256)
            and net.actor1.weight.shape == (64, 128) and net.actor2.weight.shape == (4, 64)
            and all(p.is_cuda and p.dtype == torch.float32 and p.is_contiguous() for p in net.parameters()))


def head(net, x, want_q=True, want_greedy=False):
    """pool -> conv7 -> flatten -> fc1 -> fc2 -> actor1 -> actor2 (DQNNet.py:52-63, eval mode) on the trunk's output — f32
    [B, 64, 12, 12] or [B, 64, 26, 26], or the PX16 image the weight-stationary chain ends in — in one library call
    (csrc/tron_head.hip).  Returns Q [B, 4] and / or the greedy action int8 [B]."""
    L = nat.lib()
    px = isinstance(x, PX16)                      # conv6's output as the PX16 image: the pooling reads it as it is
    B, side = x.shape[0], x.shape[-1]
    if isinstance(x, Pooled12):                   # ... or already pooled (tron_conv3x3_ws_fwd_pool12)
        dev, fn, xp, side = x.buf.device, L.tron_dqn_head_fwd_pooled, x.buf.data_ptr(), 12
    elif px:
        assert tuple(x.shape[1:]) == (64, side, side), x.shape
        dev, fn, xp = x.buf.device, L.tron_dqn_head_fwd_px16, x.buf.data_ptr()
    else:
        assert x.dtype == torch.float32 and x.is_contiguous() and tuple(x.shape[1:]) == (64, side, side), x.shape
        dev, fn, xp = x.device, L.tron_dqn_head_fwd, x.data_ptr()
    q = torch.empty(B, 4, dtype=torch.float32, device=dev) if want_q else None
    g = torch.empty(B, dtype=torch.int8, device=dev) if want_greedy else None
    if B == 0:
        return (q, g) if want_greedy else q
    ws = torch.empty(int(L.tron_dqn_head_workspace(B, side)), dtype=torch.uint8, device=dev)
    ptr = lambda t: None if t is None else t.data_ptr()
    with torch.cuda.device(dev):
        nat.check(fn(xp, B, side, net.conv7.weight.data_ptr(), net.conv7.bias.data_ptr(),
                                      net.fc1.weight.data_ptr(), net.fc1.bias.data_ptr(), net.fc2.weight.data_ptr(),
                                      net.fc2.bias.data_ptr(), net.actor1.weight.data_ptr(), net.actor1.bias.data_ptr(),
                                      net.actor2.weight.data_ptr(), net.actor2.bias.data_ptr(), ws.data_ptr(), ptr(q),
                                      ptr(g), torch.cuda.current_stream(dev).cuda_stream), "tron_dqn_head_fwd")
    return (q, g) if want_greedy else q


def pool_s2(x):
    """AvgPool2d(3, stride 2, padding 1) of f32 [B, C, S, S], S 12 or 26 (tron_pool_s2; DQNNet.py:20,52)."""
    B, C, S, _ = x.shape
    assert x.dtype == torch.float32 and x.is_contiguous() and S in (12, 26)
    y = torch.empty(B, C, S // 2, S // 2, dtype=torch.float32, device=x.device)
    if B:
        with torch.cuda.device(x.device):
            nat.check(nat.lib().tron_pool_s2(x.data_ptr(), y.data_ptr(), B * C, S, torch.cuda.current_stream(x.device).cuda_stream),
                      "tron_pool_s2")
    return y


use_gemm = _os.environ.get("TRON_HEAD_GEMM", "1") != "0"      # the training head's dense products on tron_gemm_f16x3 (0: the library)


def gemm_f16x3(a, b, bias=None, a_transposed=False, b_transposed=False, a_scale=None):
    """a @ b.T + bias on the split-f16 matrix cores (tron_gemm_f16x3), f32 in and out.  a: [M, K] (or [K, M] with
    a_transposed), b: [N, K] (or [K, N] with b_transposed).  None where the kernel does not cover the shape (the caller then
    uses the library).  a_scale: a device scalar (power of two) for a gradient operand `a`."""
    if not (use_gemm and default_math == "f16x3" and a.is_cuda and a.dtype == torch.float32 and b.dtype == torch.float32
            and a.dim() == 2 and b.dim() == 2):
        return None
    M, K = (a.shape[1], a.shape[0]) if a_transposed else a.shape
    N, Kb = (b.shape[1], b.shape[0]) if b_transposed else b.shape
    L = nat.lib()
    nbytes = int(L.tron_gemm_f16x3_workspace(M, N, K)) if Kb == K and M > 0 else 0
    if nbytes <= 0 or ((not a_transposed or not b_transposed) and K % 64 != 0):
        return None
    a, b = a.contiguous(), b.contiguous()
    out = torch.empty(M, N, dtype=torch.float32, device=a.device)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=a.device)
    with torch.cuda.device(a.device):
        nat.check(L.tron_gemm_f16x3(nat.ptr(a), int(a_transposed), nat.ptr(b), int(b_transposed), nat.ptr(None if bias is None else bias.contiguous()),
                                    nat.ptr(a_scale), nat.ptr(out), M, N, K, nat.ptr(ws), nat.stream_ptr()), "tron_gemm_f16x3")
    return out
