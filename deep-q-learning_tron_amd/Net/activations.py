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
        if not _aligned16(g):
            g = g.clone(memory_format=torch.contiguous_format)
        gx = torch.empty_like(x)
        with torch.cuda.device(x.device):
            nat.check(nat.lib().tron_mish_bwd(nat.ptr(x), nat.ptr(g), nat.ptr(gx), x.numel(), nat.stream_ptr()),
                      "tron_mish_bwd")
        return gx


def _aligned16(*tensors):
    """The HIP passes use 16-byte accesses: a contiguous view at an odd storage offset (t[1:]) is not eligible."""
    return all(t is None or (t.is_contiguous() and t.data_ptr() % 16 == 0) for t in tensors)


def mish(x):
    if x.is_cuda and x.dtype == torch.float32 and x.numel() > 0 and _aligned16(x):
        return _Mish.apply(x)
    return F.mish(x)


def bias_mish_bwd(pre, g, want_absmax=False):
    """(grad_pre, grad_bias) for out = mish(pre), pre = y + bias[c] (+ residual): one pass of csrc/tron_nn.hip.
    want_absmax: also the per-block maxima of |grad_pre| (f32 [C * 64]) that the conv gradient kernels scale by."""
    from tron import _native as nat
    N, C, H, W = pre.shape
    gp = torch.empty_like(pre)
    gb = torch.empty(C, dtype=torch.float32, device=pre.device)
    scratch = torch.empty(C * 128, dtype=torch.float32, device=pre.device)
    with torch.cuda.device(pre.device):
        nat.check(nat.lib().tron_bias_mish_bwd(nat.ptr(pre), nat.ptr(g), nat.ptr(gp), nat.ptr(gb), nat.ptr(scratch), N, C,
                                               H * W, nat.stream_ptr()), "tron_bias_mish_bwd")
    return (gp, gb, scratch[C * 64:]) if want_absmax else (gp, gb)


class _ConvBiasMishHIP(torch.autograd.Function):
    """mish(conv3x3(x) + bias (+ residual)) with forward, input gradient and weight gradient on the hand-written kernels
    (csrc/tron_conv_f16.hip, tron_conv_wgrad.hip) and activation + bias gradient in one pass of csrc/tron_nn.hip.  Shapes
    the gradient kernels do not cover (24x24 boards' weight gradient) go through aten.convolution_backward."""

    @staticmethod
    def forward(ctx, x, weight, bias, residual, grad_pre_hook=None):
        """grad_pre_hook (may be None) is called in backward with the gradient at the pre-activation: what K-FAC's statistics
        hooks of the convolution module and of its split-off bias layer take (Net/kfac.py::SplitBias, kfac.py:156-189)."""
        from Net import fused
        out, pre = fused.conv3x3_raw(x, weight, bias.reshape(-1), residual, act=True, want_pre=True)
        ctx.save_for_backward(x, weight, pre)
        ctx.has_res = residual is not None
        ctx.bias_shape = tuple(bias.shape)
        ctx.grad_pre_hook = grad_pre_hook
        ctx.grad_scope = _current_scope()
        return out

    @staticmethod
    def backward(ctx, grad_out):
        from Net import fused
        x, weight, pre = ctx.saved_tensors
        g = grad_out.contiguous()
        if not _aligned16(g):
            g = g.clone(memory_format=torch.contiguous_format)
        gp, gb, absmax = bias_mish_bwd(pre, g, want_absmax=True)
        if ctx.grad_pre_hook is not None:
            ctx.grad_pre_hook(gp)
        gx = None
        if ctx.needs_input_grad[0]:
            if weight.shape[1] in (32, 64):
                gx = fused.conv3x3_dgrad(gp, weight.detach(), absmax)
            else:                                 # conv1's planes asked for their gradient (saliency, a layer in front): the library
                gx = torch.ops.aten.convolution_backward(gp, x, weight, None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1,
                                                         [True, False, False])[0]
        gw = None
        if ctx.needs_input_grad[1] and not (_skips_weight_gradient(ctx) and weight.shape[1] >= 16):
            if fused.wgrad_supported(weight, x.shape[-1]):
                gw = fused.conv3x3_wgrad(x, gp, absmax)
            else:
                gw = torch.ops.aten.convolution_backward(gp, x, weight, None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1,
                                                         [False, True, False])[1]
        return gx, gw, (gb.reshape(ctx.bias_shape) if ctx.needs_input_grad[2] else None), (gp if ctx.has_res else None), None


class _Conv1CodesHIP(torch.autograd.Function):
    """mish(conv1(pop_up(codes)) + bias) for the learner, conv1 reading the env's / the replay ring's int8 observation
    codes directly (TRON_CONV_IN_CODES: util.pop_up's planes, util.py:11-37, are built while the kernel stages them).
    conv1 needs no input gradient; the weight gradient takes the f32 planes, which are built only in backward()."""

    @staticmethod
    def forward(ctx, codes, weight, bias, plane4):
        from Net import fused
        out, pre = fused.conv3x3_raw(codes, weight, bias, None, act=True, codes=True, plane4=plane4, want_pre=True)
        ctx.save_for_backward(codes, weight, pre)
        ctx.plane4 = plane4
        return out

    @staticmethod
    def backward(ctx, grad_out):
        from Net import fused
        from tron.vec import pop_up_planes
        codes, weight, pre = ctx.saved_tensors
        g = grad_out.contiguous()
        if not _aligned16(g):
            g = g.clone(memory_format=torch.contiguous_format)
        gp, gb, absmax = bias_mish_bwd(pre, g, want_absmax=True)
        gw = None
        if ctx.needs_input_grad[1]:
            x = pop_up_planes(codes)
            if weight.shape[1] == 4:
                x = torch.cat([x, torch.full_like(x[:, :1], ctx.plane4)], 1)
            if fused.wgrad_supported(weight, x.shape[-1]):
                gw = fused.conv3x3_wgrad(x, gp, absmax)
            else:
                gw = torch.ops.aten.convolution_backward(gp, x, weight, None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1,
                                                         [False, True, False])[1]
        return None, gw, (gb if ctx.needs_input_grad[2] else None), None


# A backward pass that exists only for K-FAC's gradient statistics (ACKTR.Brain.update) does not want the hand-written
# convolution nodes' weight gradients, and a custom Function cannot learn that from the engine: ctx.needs_input_grad is fixed
# at forward time, so `backward(inputs=...)` alone does not reach it (the library operators do learn it).  The switch is
# scoped to the GRAPH, not to the process: a Brain owns a GradScope, its forward passes run inside `with grad_scope(scope)`
# (forward runs on the calling thread: a thread-local), every node built there keeps a reference to the scope, and the
# Brain flips `scope.skip_weight_gradients` around its statistics pass.  A backward of any other graph — the other player's
# Brain, another thread — holds another scope (or none) and is unaffected.  (Backward itself runs on autograd's device
# thread, which is why the node carries the scope instead of looking one up.)
import threading as _threading


class GradScope:
    __slots__ = ("skip_weight_gradients",)

    def __init__(self):
        self.skip_weight_gradients = False


_forward_scope = _threading.local()


class grad_scope:
    """Context manager: nodes created inside belong to `scope`."""

    def __init__(self, scope):
        self.scope = scope

    def __enter__(self):
        self.prev = getattr(_forward_scope, "scope", None)
        _forward_scope.scope = self.scope
        return self.scope

    def __exit__(self, *exc):
        _forward_scope.scope = self.prev
        return False


def _current_scope():
    return getattr(_forward_scope, "scope", None)


def _skips_weight_gradient(ctx):
    scope = getattr(ctx, "grad_scope", None)
    return scope is not None and scope.skip_weight_gradients


class _Conv3x3HIP(torch.autograd.Function):
    """conv3x3(x, weight, padding=1) (+ bias) alone — no activation — on the hand-written kernels, for modules whose bias
    and activation are separate layers: the ACKTR nets after KFACOptimizer split their biases (Net/kfac.py::SplitBias),
    whose K-FAC hooks sit on the nn.Conv2d module around this node (kfac.py:156-189)."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        from Net import fused
        out = fused.conv3x3_raw(x, weight, bias, None, act=False)
        ctx.save_for_backward(x, weight)
        ctx.grad_scope = _current_scope()
        return out

    @staticmethod
    def backward(ctx, grad_out):
        from Net import fused
        x, weight = ctx.saved_tensors
        g = grad_out.contiguous()
        if not _aligned16(g):
            g = g.clone(memory_format=torch.contiguous_format)
        from tron import _native as nat                       # max |g| in one launch: the gradient's scale on its way into f16 (tron_conv3x3_dgrad)
        o4 = torch.zeros(4, dtype=torch.float32, device=g.device)
        with torch.cuda.device(g.device):
            nat.check(nat.lib().tron_absmax_pow2(nat.ptr(g), g.numel(), 16, nat.ptr(o4), nat.stream_ptr()), "tron_absmax_pow2")
        absmax = o4[1:2]
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            if weight.shape[1] in (32, 64):
                gx = fused.conv3x3_dgrad(g, weight.detach(), absmax)
            else:
                gx = torch.ops.aten.convolution_backward(g, x, weight, None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1,
                                                         [True, False, False])[0]
        if ctx.needs_input_grad[1] and not (_skips_weight_gradient(ctx) and weight.shape[1] >= 16):
            if fused.wgrad_supported(weight, x.shape[-1]):
                gw = fused.conv3x3_wgrad(x, g, absmax)
            else:                                             # (shapes the weight-gradient kernels do not cover)
                gw = torch.ops.aten.convolution_backward(g, x, weight, None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1,
                                                         [False, True, False])[1]
        if ctx.needs_input_grad[2]:
            gb = g.sum((0, 2, 3))
        return gx, gw, gb


class Conv3x3(torch.nn.Conv2d):
    """nn.Conv2d(cin, cout, 3, padding=1) whose forward runs on the hand-written kernels where they cover the shape (f32
    CUDA tensors of side 12 / 26 / 34, 32 or 64 output channels) and on the library otherwise.  Same parameters, same
    state_dict keys, still an nn.Conv2d for whoever looks for one (KFACOptimizer's hooks and factor shapes)."""

    def forward(self, x):
        from Net import fused
        if (x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and x.shape[-1] == x.shape[-2] and x.shape[0] > 0
                and x.shape[1] == self.in_channels and fused.supported(self, x.shape[-1]) and _aligned16(x)
                and (self.in_channels in (3, 4, 32, 64)) and not (self.in_channels in (3, 4) and self.out_channels != 32)):
            return _Conv3x3HIP.apply(x.contiguous(), self.weight, self.bias)
        return super().forward(x)


class _TrunkHIP(torch.autograd.Function):
    """conv1 .. conv6 with their two residual connections (DQNNet.py:33-50) as ONE autograd node.  Forward is the six
    layer kernels `_ConvBiasMishHIP` runs.  Backward needs the node to see the whole chain: between two layers autograd
    would run the convolution's input gradient, an elementwise add where a residual connection delivers a second
    gradient, and the activation's backward + bias sum — three passes over the tensor; here the input-gradient kernel's
    epilogue adds the residual term, multiplies by mish'(pre) of the layer below and leaves that layer's bias sums and
    the scale of what it wrote (tron_conv3x3_dgrad_mish), so a gradient tensor is written once and read only by the
    kernels that contract it.  x: f32 planes [B, C, S, S] or the env's int8 codes [B, S, S]."""

    @staticmethod
    def forward(ctx, x, plane4, *wb):
        from Net import fused
        w, b = wb[0::2], wb[1::2]
        codes = x.dtype == torch.int8
        ws = fused.split_weights(list(w))                            # all six layers' split weights in one launch
        a1, z1 = fused.conv3x3_raw(x, w[0], b[0], None, act=True, codes=codes, plane4=plane4, want_pre=True, presplit=ws[0])
        a2, z2 = fused.conv3x3_raw(a1, w[1], b[1], None, act=True, want_pre=True, presplit=ws[1])
        a3, z3 = fused.conv3x3_raw(a2, w[2], b[2], a1, act=True, want_pre=True, presplit=ws[2])
        a4, z4 = fused.conv3x3_raw(a3, w[3], b[3], None, act=True, want_pre=True, presplit=ws[3])
        a5, z5 = fused.conv3x3_raw(a4, w[4], b[4], None, act=True, want_pre=True, presplit=ws[4])
        a6, z6 = fused.conv3x3_raw(a5, w[5], b[5], a4, act=True, want_pre=True, presplit=ws[5])
        ctx.save_for_backward(x, a1, a2, a3, a4, a5, z1, z2, z3, z4, z5, z6, *w)
        ctx.plane4 = plane4
        return a6

    @staticmethod
    def backward(ctx, grad_out):
        from Net import fused
        x, a1, a2, a3, a4, a5, z1, z2, z3, z4, z5, z6, *w = ctx.saved_tensors
        g = grad_out.contiguous()
        if not _aligned16(g):
            g = g.clone(memory_format=torch.contiguous_format)
        side = g.shape[-1]
        need = ctx.needs_input_grad                                 # (x, plane4, w1, b1, ..., w6, b6)

        def wgrad(layer, inp, gp, am):
            if not need[2 + 2 * layer]:
                return None
            if fused.wgrad_supported(w[layer], side):
                return fused.conv3x3_wgrad(inp, gp, am)
            return torch.ops.aten.convolution_backward(gp, inp, w[layer], None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1,
                                                       [False, True, False])[1]

        gb = [None] * 6
        gw = [None] * 6
        gp6, gb[5], am = bias_mish_bwd(z6, g, want_absmax=True)
        del g
        gw[5] = wgrad(5, a5, gp6, am)
        gp5, gb[4], am5 = fused.conv3x3_dgrad_mish(gp6, w[5], am, z5)
        gw[4] = wgrad(4, a4, gp5, am5)
        gp4, gb[3], am4 = fused.conv3x3_dgrad_mish(gp5, w[4], am5, z4, extra=gp6)    # a4 also feeds conv6's residual
        del gp5, gp6
        gw[3] = wgrad(3, a3, gp4, am4)
        gp3, gb[2], am3 = fused.conv3x3_dgrad_mish(gp4, w[3], am4, z3)
        del gp4
        gw[2] = wgrad(2, a2, gp3, am3)
        gp2, gb[1], am2 = fused.conv3x3_dgrad_mish(gp3, w[2], am3, z2)
        gw[1] = wgrad(1, a1, gp2, am2)
        gp1, gb[0], am1 = fused.conv3x3_dgrad_mish(gp2, w[1], am2, z1, extra=gp3)    # a1 also feeds conv3's residual
        del gp2, gp3
        gx = None
        planes = x
        if x.dtype == torch.int8:
            if need[2]:
                from tron.vec import pop_up_planes
                planes = pop_up_planes(x)
                if w[0].shape[1] == 4:
                    planes = torch.cat([planes, torch.full_like(planes[:, :1], ctx.plane4)], 1)
        elif need[0]:                                               # the planes asked for their gradient (saliency): the library
            gx = torch.ops.aten.convolution_backward(gp1, x, w[0], None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1,
                                                     [True, False, False])[0]
        gw[0] = wgrad(0, planes, gp1, am1)
        grads = [gx, None]
        for k in range(6):
            grads += [gw[k], gb[k] if need[3 + 2 * k] else None]
        return tuple(grads)


def _trunk_px_forward(codes, plane4, w, b, last_f32, pooled12=False):
    """conv1 .. conv6 on the weight-stationary chain with the pre-activations kept (csrc/tron_conv_ws_train.hip).  Returns conv6's
    output (f32 NCHW if last_f32, else PX16; pooled12 at 12x12: its AvgPool2d(3, 2, 1) as f32 [B, 64 * 36] instead — conv6 and the
    pooling are then one launch and conv6's output is never stored) and the tensors the backward needs."""
    from Net import fused
    frag = fused._split_jobs(list(w[1:6]), "tron_conv3x3_ws_split_weights", False)
    a1, z1 = fused.conv1_px16_train(codes, w[0], b[0], plane4)
    a2, z2 = fused.conv_ws_train(a1, 32, frag[0], b[1])
    a3, z3 = fused.conv_ws_train(a2, 32, frag[1], b[2], residual=a1)
    a4, z4 = fused.conv_ws_train(a3, 64, frag[2], b[3])
    a5, z5 = fused.conv_ws_train(a4, 64, frag[3], b[4])
    if pooled12 and fused.use_pool_fused_train and codes.shape[-1] == 12 and codes.shape[0] > 0 and not last_f32:
        out, z6 = fused.conv_ws_train_pool12(a5, frag[4], b[5], a4)
    else:
        out, z6 = fused.conv_ws_train(a5, 64, frag[4], b[5], residual=a4, want_f32=last_f32)
    return out, (codes, a1.buf, a2.buf, a3.buf, a4.buf, a5.buf, z1.buf, z2.buf, z3.buf, z4.buf, z5.buf, z6.buf)


def _trunk_px_backward(saved, w, plane4, gp6, gb6, need):
    """The gradient chain below conv6's activation backward: gp6 = the gradient image at conv6's pre-activation, gb6 conv6's bias
    gradient.  need[k]: conv(k+1)'s weight gradient is wanted.  Returns (gw[6], gb[6])."""
    from Net import fused
    codes, a1, a2, a3, a4, a5, z1, z2, z3, z4, z5, z6 = saved
    B, S = gp6.shape[0], gp6.shape[-1]
    px = lambda buf, c: fused._px((B, c, S, S), buf)
    rot, wn = fused._split_jobs(list(w[1:6]), "tron_conv3x3_ws_split_weights_bwd", True)      # conv2 .. conv6, one launch
    gb, gw = [None] * 6, [None] * 6
    gb[5] = gb6
    gw[5] = fused.conv3x3_wgrad_px(px(a5, 64), gp6) if need[5] else None
    gp5, _, gb[4] = fused.conv_ws_dgrad(gp6, 64, rot[4], wn[4:5], px(z5, 64))
    gw[4] = fused.conv3x3_wgrad_px(px(a4, 64), gp5) if need[4] else None
    gp4, _, gb[3] = fused.conv_ws_dgrad(gp5, 64, rot[3], wn[3:4], px(z4, 64), extra=gp6)      # a4 also feeds conv6's residual
    del gp5, gp6
    gw[3] = fused.conv3x3_wgrad_px(px(a3, 32), gp4) if need[3] else None
    gp3, _, gb[2] = fused.conv_ws_dgrad(gp4, 32, rot[2], wn[2:3], px(z3, 32))
    del gp4
    gw[2] = fused.conv3x3_wgrad_px(px(a2, 32), gp3) if need[2] else None
    gp2, _, gb[1] = fused.conv_ws_dgrad(gp3, 32, rot[1], wn[1:2], px(z2, 32))
    gw[1] = fused.conv3x3_wgrad_px(px(a1, 32), gp2) if need[1] else None
    gp1, _, gb[0] = fused.conv_ws_dgrad(gp2, 32, rot[0], wn[0:1], px(z1, 32), extra=gp3)    # a1 also feeds conv3's residual
    del gp2, gp3
    if need[0]:
        gw[0] = fused.conv1_wgrad_px(codes, gp1, w[0].shape[1], plane4)       # from the codes themselves: no f32 planes, no f32 gradient
    return gw, gb


class _TrunkPX(torch.autograd.Function):
    """conv1 .. conv6 (DQNNet.py:33-50) from the env's int8 codes as ONE autograd node on the weight-stationary design
    (csrc/tron_conv_ws_train.hip): the forward is the gradient-free chain's kernel keeping every pre-activation as a PX16 image,
    the backward runs each input gradient as that same kernel on the rotated weights — residual gradient, mish' of the layer
    below, its bias sums and the next scale in the epilogue — and each weight gradient straight from the two PX16 images
    (transposed LDS reads).  No f32 NCHW tensor exists between conv1's output and conv6's output; this node hands conv6's f32
    output to whatever follows (`_BodyPX` also takes the head's pooling and conv7 in); conv1's weight gradient (3 or 4 planes ->
    32: plain f32 FMAs) reads an f32 gradient the last input-gradient launch also writes."""

    @staticmethod
    def forward(ctx, codes, plane4, *wb):
        w, b = wb[0::2], wb[1::2]
        out, saved = _trunk_px_forward(codes, plane4, w, b, True)
        ctx.save_for_backward(*saved, *w)
        ctx.plane4 = plane4
        return out

    @staticmethod
    def backward(ctx, grad_out):
        from Net import fused
        *saved, w1, w2, w3, w4, w5, w6 = ctx.saved_tensors
        g = grad_out.contiguous()
        if not _aligned16(g):
            g = g.clone(memory_format=torch.contiguous_format)
        B, S = g.shape[0], g.shape[-1]
        need = ctx.needs_input_grad                                 # (codes, plane4, w1, b1, ..., w6, b6)
        gp6, gb6 = fused.grad_px_from_f32(g, fused._px((B, 64, S, S), saved[11]))
        del g
        gw, gb = _trunk_px_backward(saved, (w1, w2, w3, w4, w5, w6), ctx.plane4, gp6, gb6, [need[2 + 2 * k] for k in range(6)])
        grads = [None, None]
        for k in range(6):
            grads += [gw[k], gb[k] if need[3 + 2 * k] else None]
        return tuple(grads)


class TrunkHooks:
    """K-FAC's statistics hooks of the six trunk layers, fed by hand by `_ACTrunkPX` (the modules themselves are not called):
    per layer the convolution module and its split-off bias layer (Net/kfac.py::SplitBias) — they would have seen the layer's
    input on the way in and the gradient at its pre-activation on the way back (kfac.py:156-189)."""

    def __init__(self, convs):
        self.convs = list(convs)

    def _opt(self, k):
        return getattr(self.convs[k].add_bias, "_kfac", None)

    def inputs(self, k, a, batch_like):
        opt = self._opt(k)
        if opt is not None:
            opt._save_input(self.convs[k].module, (a,))
            opt._save_input(self.convs[k].add_bias, (batch_like,))        # (an AddBias's input factor takes the batch size only)

    def grads(self, k, g):
        opt = self._opt(k)
        if opt is not None:
            opt._save_grad_output(self.convs[k].module, None, (g,))
            opt._save_grad_output(self.convs[k].add_bias, None, (g,))

    def wants_grads(self):
        return any(self._opt(k) is not None and self._opt(k).acc_stats for k in range(len(self.convs)))


class _ACTrunkPX(torch.autograd.Function):
    """conv1 .. conv6 with their two residual connections — the trunk every actor-critic net shares (ACNet.py:97-111) — from f32
    planes, as ONE autograd node on the weight-stationary kernels (csrc/tron_conv_ws_train.hip; 12x12, 26x26 and 34x34): conv1 on
    the chunked kernel (3 or 4 planes in), its output and pre-activation turned into PX16 images once, conv2 .. conv6 as the
    learner's chain (`_BodyPX`), the backward as that chain with every gradient also written as f32 for K-FAC's gradient
    factors.  hooks (TrunkHooks or None): K-FAC's statistics, fed by hand."""

    @staticmethod
    def forward(ctx, x, hooks, *wb):
        from Net import fused
        w, b = wb[0::2], [t.reshape(-1) for t in wb[1::2]]
        y1, z1f = fused.conv3x3_raw(x, w[0], b[0], None, act=True, want_pre=True)
        a1, z1 = fused.PX16.from_f32(y1), fused.PX16.from_f32(z1f)
        del y1, z1f
        frag = fused._split_jobs(list(w[1:6]), "tron_conv3x3_ws_split_weights", False)
        a2, z2 = fused.conv_ws_train(a1, 32, frag[0], b[1])
        a3, z3 = fused.conv_ws_train(a2, 32, frag[1], b[2], residual=a1)
        a4, z4 = fused.conv_ws_train(a3, 64, frag[2], b[3])
        a5, z5 = fused.conv_ws_train(a4, 64, frag[3], b[4])
        out, z6 = fused.conv_ws_train(a5, 64, frag[4], b[5], residual=a4, want_f32=True)
        if hooks is not None:
            like = x.new_empty((x.shape[0], 0))
            with torch.enable_grad():                               # (the hooks skip gradient-free forwards; a Function's forward runs as one)
                for k, a in enumerate((x, a1, a2, a3, a4, a5)):
                    hooks.inputs(k, a, like)
        ctx.save_for_backward(x, a1.buf, a2.buf, a3.buf, a4.buf, a5.buf, z1.buf, z2.buf, z3.buf, z4.buf, z5.buf, z6.buf, *w)
        ctx.hooks, ctx.bias_shapes, ctx.grad_scope = hooks, [tuple(t.shape) for t in wb[1::2]], _current_scope()
        return out

    @staticmethod
    def backward(ctx, grad_out):
        from Net import fused
        x, a1, a2, a3, a4, a5, z1, z2, z3, z4, z5, z6, *w = ctx.saved_tensors
        g = grad_out.contiguous()
        if not _aligned16(g):
            g = g.clone(memory_format=torch.contiguous_format)
        B, S = g.shape[0], g.shape[-1]
        px = lambda buf, c: fused._px((B, c, S, S), buf)
        need = ctx.needs_input_grad                                 # (x, hooks, w1, b1, ..., w6, b6)
        skip = _skips_weight_gradient(ctx)                          # the sampled-Fisher pass: statistics only
        want = lambda k: need[2 + 2 * k] and not skip
        hooks = ctx.hooks
        stats = hooks is not None and hooks.wants_grads()
        gp6, gb6 = fused.grad_px_from_f32(g, px(z6, 64))
        del g
        if stats:
            hooks.grads(5, gp6.float())
        rot, wn = fused._split_jobs(list(w[1:6]), "tron_conv3x3_ws_split_weights_bwd", True)
        gw, gb = [None] * 6, [None] * 6
        gb[5] = gb6
        gw[5] = fused.conv3x3_wgrad_px(px(a5, 64), gp6) if want(5) else None
        gp5, f5, gb[4] = fused.conv_ws_dgrad(gp6, 64, rot[4], wn[4:5], px(z5, 64), want_f32=stats)
        if stats:
            hooks.grads(4, f5)
        del f5
        gw[4] = fused.conv3x3_wgrad_px(px(a4, 64), gp5) if want(4) else None
        gp4, f4, gb[3] = fused.conv_ws_dgrad(gp5, 64, rot[3], wn[3:4], px(z4, 64), extra=gp6, want_f32=stats)   # a4 also feeds conv6's residual
        del gp5, gp6
        if stats:
            hooks.grads(3, f4)
        del f4
        gw[3] = fused.conv3x3_wgrad_px(px(a3, 32), gp4) if want(3) else None
        gp3, f3, gb[2] = fused.conv_ws_dgrad(gp4, 32, rot[2], wn[2:3], px(z3, 32), want_f32=stats)
        del gp4
        if stats:
            hooks.grads(2, f3)
        del f3
        gw[2] = fused.conv3x3_wgrad_px(px(a2, 32), gp3) if want(2) else None
        gp2, f2, gb[1] = fused.conv_ws_dgrad(gp3, 32, rot[1], wn[1:2], px(z2, 32), want_f32=stats)
        if stats:
            hooks.grads(1, f2)
        del f2
        gw[1] = fused.conv3x3_wgrad_px(px(a1, 32), gp2) if want(1) else None
        _, f1, gb[0] = fused.conv_ws_dgrad(gp2, 32, rot[0], wn[0:1], px(z1, 32), extra=gp3, want_px=False, want_f32=True)   # a1 also feeds conv3's residual
        del gp2, gp3
        if stats:
            hooks.grads(0, f1)
        if need[2]:                                                 # conv1's weight (3 or 4 planes in: a cheap product, also in the statistics pass)
            if fused.wgrad_supported(w[0], S):
                gw[0] = fused.conv3x3_wgrad(x, f1)
            else:
                gw[0] = torch.ops.aten.convolution_backward(f1, x, w[0], None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1, [False, True, False])[1]
        gx = None
        if need[0]:                                                 # the planes asked for their gradient (saliency): the library
            gx = torch.ops.aten.convolution_backward(f1, x, w[0], None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1, [True, False, False])[0]
        grads = [gx, None]
        for k in range(6):
            grads += [gw[k], gb[k].reshape(ctx.bias_shapes[k]) if need[3 + 2 * k] else None]
        return tuple(grads)


def ac_trunk_infer(x, weights, biases):
    """conv1 .. conv6 of the actor-critic trunk, gradient-free (the rollouts' acting, ACKTR.py:285-289), on the weight-stationary
    chain: conv1 on the chunked kernel, one f32 -> PX16 conversion, conv2 .. conv6 with their weights in registers and the
    activations as PX16 images, conv6 writing f32 for the pooling.  The caller checked ac_trunk_px_supported(..., need_grad=False)."""
    from Net import fused
    b = [t.reshape(-1) for t in biases]
    a1 = fused.PX16.from_f32(fused.conv3x3_raw(x, weights[0], b[0], None, act=True))
    frag = fused._split_jobs(list(weights[1:6]), "tron_conv3x3_ws_split_weights", False)
    a2 = fused.conv_ws_infer(a1, 32, frag[0], b[1])
    a3 = fused.conv_ws_infer(a2, 32, frag[1], b[2], residual=a1)
    del a1, a2
    a4 = fused.conv_ws_infer(a3, 64, frag[2], b[3])
    del a3
    a5 = fused.conv_ws_infer(a4, 64, frag[3], b[4])
    return fused.conv_ws_infer(a5, 64, frag[4], b[5], residual=a4, want_f32=True)


def ac_trunk_px_supported(x, weights, need_grad=True):
    """Can `_ACTrunkPX` (need_grad) / `ac_trunk_infer` run conv1 .. conv6 (weights: the six convolution weights) on the f32 planes x?"""
    from Net import fused
    import os
    if os.environ.get("TRON_AC_TRUNK_PX", "1") == "0" or not fused.use_trunk_px:
        return False
    if not (x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and x.shape[0] > 0 and x.shape[-1] == x.shape[-2] and x.shape[-1] in (12, 26, 34)
            and x.shape[1] in (3, 4) and _aligned16(x) and torch.is_grad_enabled() == need_grad and fused.default_math == "f16x3"
            and bias_mish_supported(x)):                     # (the switch the tests use to get the module-by-module graph)
        return False
    chans = [(w.shape[1], w.shape[0]) for w in weights]
    return (chans == [(x.shape[1], 32), (32, 32), (32, 32), (32, 64), (64, 64), (64, 64)]
            and all(tuple(w.shape[2:]) == (3, 3) and w.is_cuda and w.dtype == torch.float32 and w.is_contiguous() for w in weights))


class _BodyPX(torch.autograd.Function):
    """The whole convolutional body of the DQN net — conv1 .. conv6, AvgPool2d(3, 2, 1), conv7, mish, flatten (DQNNet.py:33-55) —
    from the env's int8 codes as one node: `_TrunkPX` with the head's first two layers taken in, so that conv6's output goes into
    the pooling as the PX16 image it is, and on the way back the pooling's backward, conv6's activation backward, its bias sums
    and the gradient image's scale are ONE pass over the pooled gradient (tron_px16_grad_from_pooled) — the f32 planes of the
    trunk's output and of their gradient are never written.  12x12: conv7 on its dense form (`_PoolConv7`'s GEMMs); 26x26:
    tron_pool_conv7_fwd_px16 / tron_pool_conv7_bwd_pooled (`_PoolConv7CL`'s kernels)."""

    @staticmethod
    def forward(ctx, codes, plane4, *wb):
        from tron import _native as nat
        from Net import fused
        L = nat.lib()
        w, b = wb[0::2], wb[1::2]
        a6, saved = _trunk_px_forward(codes, plane4, w, b, False, pooled12=True)
        B, S = codes.shape[0], codes.shape[-1]
        dev = codes.device
        st = nat.stream_ptr()
        with torch.cuda.device(dev):
            if S == 12:
                if torch.is_tensor(a6):                                   # conv6 and the pooling were one launch
                    pooled = a6
                else:
                    pooled = torch.empty(B, 64 * 36, dtype=torch.float32, device=dev)
                    nat.check(L.tron_pool12_px16(nat.ptr(a6.buf), nat.ptr(pooled), B, st), "tron_pool12_px16")
                dense = torch.empty(64 * 9, 64 * 36, dtype=torch.float32, device=dev)
                nat.check(L.tron_conv7_dense(nat.ptr(w[6]), nat.ptr(dense), 64, 64, 0, st), "tron_conv7_dense")
                pre = fused.gemm_f16x3(pooled, dense, b[6].repeat_interleave(9))
                if pre is None:
                    pre = torch.addmm(b[6].repeat_interleave(9), pooled, dense.t())
                y = torch.empty_like(pre)
                nat.check(L.tron_mish_fwd(nat.ptr(pre), nat.ptr(y), pre.numel(), st), "tron_mish_fwd")
                extra = (pooled, dense, pre)
            else:
                o = (S // 2 + 1) // 2
                keep = torch.empty(int(L.tron_pool_conv7_saved_bytes(B, S)), dtype=torch.uint8, device=dev)
                ws = torch.empty(int(L.tron_pool_conv7_workspace(B, S)), dtype=torch.uint8, device=dev)
                pre = torch.empty(B, o * o, 64, dtype=torch.float32, device=dev)
                y = torch.empty(B, 64 * o * o, dtype=torch.float32, device=dev)
                nat.check(L.tron_pool_conv7_fwd_px16(nat.ptr(a6.buf), B, S, nat.ptr(w[6]), nat.ptr(b[6]), nat.ptr(keep), nat.ptr(pre), nat.ptr(y),
                                                     nat.ptr(ws), st), "tron_pool_conv7_fwd_px16")
                extra = (keep, pre)
        del a6
        ctx.save_for_backward(*saved, *w, *extra)
        ctx.plane4, ctx.side = plane4, S
        return y

    @staticmethod
    def backward(ctx, grad_out):
        from tron import _native as nat
        from Net import fused
        L = nat.lib()
        t = ctx.saved_tensors
        saved, w, extra = t[:12], t[12:19], t[19:]
        S = ctx.side
        g = grad_out.contiguous()
        if not _aligned16(g):
            g = g.clone(memory_format=torch.contiguous_format)
        B = g.shape[0]
        dev = g.device
        need = ctx.needs_input_grad                                 # (codes, plane4, w1, b1, ..., w7, b7)
        st = nat.stream_ptr()
        gw7 = gb7 = None
        with torch.cuda.device(dev):
            if S == 12:
                pooled, dense, pre = extra
                gp = torch.empty_like(pre)
                nat.check(L.tron_mish_bwd(nat.ptr(pre), nat.ptr(g), nat.ptr(gp), pre.numel(), st), "tron_mish_bwd")
                from Net.kfac import _pow2_scale
                sc = _pow2_scale(gp) if fused.use_gemm else None
                gpool = fused.gemm_f16x3(gp, dense, b_transposed=True, a_scale=sc)           # [B, 64 * 36]: the pooled planes' gradient
                if gpool is None:
                    gpool = gp @ dense
                if need[14]:
                    gdense = fused.gemm_f16x3(gp, pooled, a_transposed=True, b_transposed=True, a_scale=sc)
                    if gdense is None:
                        gdense = gp.t() @ pooled
                    gw7 = torch.empty(64, 64, 7, 7, dtype=torch.float32, device=dev)
                    nat.check(L.tron_conv7_dense(nat.ptr(gdense), nat.ptr(gw7), 64, 64, 1, st), "tron_conv7_dense")
                if need[15]:
                    gb7 = gp.view(B, 64, 9).sum((0, 2))
                channels_last = 0
            else:
                keep, pre = extra
                gpool = torch.empty(B, (S // 2) ** 2, 64, dtype=torch.float32, device=dev)
                gw7 = torch.empty_like(w[6]) if need[14] else None
                gb7 = torch.empty(64, dtype=torch.float32, device=dev) if need[15] else None
                ws = torch.empty(int(L.tron_pool_conv7_workspace(B, S)), dtype=torch.uint8, device=dev)
                nat.check(L.tron_pool_conv7_bwd_pooled(nat.ptr(g), nat.ptr(pre), nat.ptr(keep), nat.ptr(w[6]), B, S, nat.ptr(gpool), nat.ptr(gw7),
                                                       nat.ptr(gb7), nat.ptr(ws), st), "tron_pool_conv7_bwd_pooled")
                channels_last = 1
            # pooling backward + conv6's activation backward + bias sums + the gradient image, one pass over the pooled gradient
            sc4 = torch.zeros(4, dtype=torch.float32, device=dev)
            gp6 = fused.GradPX(B, 64, S, dev)
            gb6 = torch.empty(64, dtype=torch.float32, device=dev)
            gws = torch.empty(int(L.tron_px16_grad_workspace(B, 64)), dtype=torch.uint8, device=dev)
            nat.check(L.tron_absmax_pow2(nat.ptr(gpool), gpool.numel(), 15, nat.ptr(sc4), st), "tron_absmax_pow2")
            nat.check(L.tron_px16_grad_from_pooled(nat.ptr(gpool), channels_last, nat.ptr(saved[11]), nat.ptr(sc4), B, 64, S, nat.ptr(gp6.buf),
                                                   nat.ptr(gp6.info), nat.ptr(gb6), nat.ptr(gws), st), "tron_px16_grad_from_pooled")
        del g, gpool
        gw, gb = _trunk_px_backward(saved, w[:6], ctx.plane4, gp6, gb6, [need[2 + 2 * k] for k in range(6)])
        grads = [None, None]
        for k in range(6):
            grads += [gw[k], gb[k] if need[3 + 2 * k] else None]
        return tuple(grads + [gw7, gb7])


def trunk_px_supported(net, x):
    """Can `_TrunkPX` run conv1..conv6 of `net` on x?  int8 codes [B, S, S] at 12x12 / 26x26, the DQN trunk's channel plan."""
    from Net import fused
    return (fused.use_trunk_px and torch.is_tensor(x) and x.is_cuda and x.dtype == torch.int8 and x.dim() == 3 and x.numel() > 0
            and fused.default_math == "f16x3" and fused.ws_supported(net, x.shape[-1]) and x.shape[-2] == x.shape[-1]
            and [(c.in_channels, c.out_channels) for c in (net.conv2, net.conv3, net.conv4, net.conv5, net.conv6)]
            == [(32, 32), (32, 32), (32, 64), (64, 64), (64, 64)])


def trunk_supported(net, x):
    """Can `_TrunkHIP` run conv1..conv6 of `net` on x (f32 planes [B, C, S, S] or int8 codes [B, S, S])?"""
    from Net import fused
    if not (torch.is_tensor(x) and x.is_cuda and fused.default_math == "f16x3" and x.numel() > 0):
        return False
    side = x.shape[-1]
    convs = [net.conv1, net.conv2, net.conv3, net.conv4, net.conv5, net.conv6]
    if x.dtype == torch.int8:
        if x.dim() != 3 or x.shape[-2] != side or net.conv1.in_channels not in (3, 4):
            return False
    elif not (x.dtype == torch.float32 and x.dim() == 4 and x.shape[-2] == side and x.shape[1] == net.conv1.in_channels
              and net.conv1.in_channels in (3, 4) and _aligned16(x)):
        return False
    chans = [(c.in_channels, c.out_channels) for c in convs[1:]]
    return (chans == [(32, 32), (32, 32), (32, 64), (64, 64), (64, 64)] and net.conv1.out_channels == 32
            and all(c.bias is not None and fused.supported(c, side) for c in convs)
            and all(fused.dgrad_mish_supported(c.weight, side) for c in convs[1:]))


def body_px_supported(net, x):
    """Can `_BodyPX` run conv1 .. conv7 of `net` on the codes x?  `_TrunkPX`'s shapes, the reference's pooling and conv7 (64 -> 64,
    7x7 / 2 / 3: DQNNet.py:20-22) and mish as the activation."""
    from Net import fused
    pool, conv = getattr(net, "pool", None), getattr(net, "conv7", None)
    return (trunk_px_supported(net, x) and _use_pool_conv7_cl and isinstance(pool, torch.nn.AvgPool2d) and pool.kernel_size == 3
            and pool.stride == 2 and pool.padding == 1 and pool.count_include_pad and not pool.ceil_mode and pool.divisor_override is None
            and isinstance(conv, torch.nn.Conv2d) and conv.kernel_size == (7, 7) and conv.stride == (2, 2) and conv.padding == (3, 3)
            and conv.dilation == (1, 1) and conv.groups == 1 and conv.bias is not None and conv.in_channels == 64 and conv.out_channels == 64
            and conv.weight.is_contiguous() and conv.weight.dtype == torch.float32 and conv.weight.is_cuda
            and _os.environ.get("TRON_BODY_PX", "1") != "0")


def body_mish(net, codes, plane4=0.0):
    """mish(conv7(pool(trunk(codes)))) flattened [B, 64 * O * O] (DQNNet.py:33-55) as one node (`_BodyPX`)."""
    wb = []
    for c in (net.conv1, net.conv2, net.conv3, net.conv4, net.conv5, net.conv6, net.conv7):
        wb += [c.weight, c.bias]
    return _BodyPX.apply(codes.contiguous(), float(plane4), *wb)


def trunk_mish(net, x, plane4=0.0):
    """mish-activated conv1..conv6 of a DQN `Net` (DQNNet.py:33-50) with autograd, as one node (`_TrunkHIP`)."""
    wb = []
    for c in (net.conv1, net.conv2, net.conv3, net.conv4, net.conv5, net.conv6):
        wb += [c.weight, c.bias]
    if trunk_px_supported(net, x):
        return _TrunkPX.apply(x.contiguous(), float(plane4), *wb)
    return _TrunkHIP.apply(x.contiguous(), float(plane4), *wb)


def conv1_codes_mish(conv, codes, plane4=0.0):
    """mish(conv1(planes of `codes`)) with autograd for weight and bias: codes int8 [B, S, S] on the device."""
    from Net import fused
    if not (codes.is_cuda and codes.dtype == torch.int8 and codes.dim() == 3 and codes.shape[-1] == codes.shape[-2]
            and fused.supported(conv, codes.shape[-1]) and conv.in_channels in (3, 4) and conv.bias is not None):
        raise TypeError("conv1_codes_mish: int8 codes [B, S, S] on the device and a conv1 the HIP kernels cover")
    return _Conv1CodesHIP.apply(codes.contiguous(), conv.weight, conv.bias, float(plane4))


class _BiasMish(torch.autograd.Function):
    """mish(y + bias[c] (+ residual)) for a bias-free convolution output y [N, C, H, W].  grad_pre_hook (may be None) is
    called in backward with the gradient at the pre-activation — where K-FAC's statistics hook of a split-off bias layer
    (Net/kfac.py::AddBias, kfac.py:156-189) would have seen it had bias, residual and activation been separate modules."""

    @staticmethod
    def forward(ctx, y, bias, residual, grad_pre_hook=None):
        from tron import _native as nat
        pre = y.contiguous()                      # overwritten with y + bias (+ residual): saved for backward
        res = None if residual is None else residual.contiguous()
        out = torch.empty_like(pre)
        N, C, H, W = pre.shape
        b = bias.reshape(-1).contiguous()
        with torch.cuda.device(pre.device):
            nat.check(nat.lib().tron_bias_mish_fwd(nat.ptr(pre), nat.ptr(b), nat.ptr(res), nat.ptr(out), N, C, H * W,
                                                   nat.stream_ptr()), "tron_bias_mish_fwd")
        ctx.save_for_backward(pre)
        ctx.has_res = residual is not None
        ctx.bias_shape = tuple(bias.shape)
        ctx.grad_pre_hook = grad_pre_hook
        return out

    @staticmethod
    def backward(ctx, grad_out):
        from tron import _native as nat
        (pre,) = ctx.saved_tensors
        g = grad_out.contiguous()
        if not _aligned16(g):
            g = g.clone(memory_format=torch.contiguous_format)
        gp, gb = bias_mish_bwd(pre, g)
        if ctx.grad_pre_hook is not None:
            ctx.grad_pre_hook(gp)
        return gp, gb.reshape(ctx.bias_shape), (gp if ctx.has_res else None), None


def bias_mish_supported(y, residual=None):
    """tron_bias_mish_fwd / _bwd's shapes: f32 CUDA [N, C, H, W] with H W a multiple of 4, 16-byte aligned."""
    return (y.is_cuda and y.dtype == torch.float32 and y.dim() == 4 and (y.shape[2] * y.shape[3]) % 4 == 0 and 0 < y.numel() < 2 ** 32
            and _aligned16(y, residual) and (residual is None or (residual.shape == y.shape and residual.dtype == torch.float32)))


def bias_mish(y, bias, residual=None, grad_pre_hook=None):
    """mish(y + bias (+ residual)) as one pass each way (bias [C] or [C, 1]); the caller checked bias_mish_supported."""
    return _BiasMish.apply(y, bias, residual, grad_pre_hook)


class _PoolConv7(torch.autograd.Function):
    """mish(conv7(avg_pool(x))) flattened, for 12x12 planes (Net/DQNNet.py:52-55): on the 6x6 pooled planes the 7x7 /
    stride 2 / pad 3 convolution is the dense map [Ci*36] -> [Co*9] (most taps fall on padding), so forward, input
    gradient and weight gradient are three plain f32 GEMMs on the matrix csrc/tron_head.hip builds from the weight
    (tron_conv7_dense) and folds its gradient back into; pooling forward / backward are tron_pool12.  Replaces MIOpen's
    conv7 kernels + their NCHW<->NHWC transposes + torch's avg_pool2d backward (1.1 ms -> 0.45 ms at 4 096 samples)."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        from tron import _native as nat
        L = nat.lib()
        B, C = x.shape[0], x.shape[1]
        Co = weight.shape[0]
        st = nat.stream_ptr()
        with torch.cuda.device(x.device):
            pooled = torch.empty(B, C * 36, dtype=torch.float32, device=x.device)
            nat.check(L.tron_pool12(nat.ptr(x), nat.ptr(pooled), B * C, 0, st), "tron_pool12")
            dense = torch.empty(Co * 9, C * 36, dtype=torch.float32, device=x.device)
            nat.check(L.tron_conv7_dense(nat.ptr(weight), nat.ptr(dense), Co, C, 0, st), "tron_conv7_dense")
            from Net import fused
            pre = fused.gemm_f16x3(pooled, dense, bias.repeat_interleave(9))            # [B, Ci*36] x [Co*9, Ci*36]^T
            if pre is None:
                pre = torch.addmm(bias.repeat_interleave(9), pooled, dense.t())
            out = torch.empty_like(pre)
            nat.check(L.tron_mish_fwd(nat.ptr(pre), nat.ptr(out), pre.numel(), st), "tron_mish_fwd")
        ctx.save_for_backward(pooled, dense, pre)
        ctx.shape = tuple(x.shape)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        from tron import _native as nat
        L = nat.lib()
        pooled, dense, pre = ctx.saved_tensors
        B, C = ctx.shape[0], ctx.shape[1]
        Co = dense.shape[0] // 9
        g = grad_out.contiguous()
        if not _aligned16(g):
            g = g.clone(memory_format=torch.contiguous_format)
        st = nat.stream_ptr()
        gx = gw = gb = None
        with torch.cuda.device(pre.device):
            gp = torch.empty_like(pre)
            nat.check(L.tron_mish_bwd(nat.ptr(pre), nat.ptr(g), nat.ptr(gp), pre.numel(), st), "tron_mish_bwd")
            from Net import fused
            from Net.kfac import _pow2_scale
            sc = _pow2_scale(gp) if fused.use_gemm else None                # the gradient's scale on its way into f16 (a device scalar)
            if ctx.needs_input_grad[0]:
                gpool = fused.gemm_f16x3(gp, dense, b_transposed=True, a_scale=sc)       # gp [B, Co*9] x dense [Co*9, Ci*36]
                if gpool is None:
                    gpool = gp @ dense
                gx = torch.empty(ctx.shape, dtype=torch.float32, device=pre.device)
                nat.check(L.tron_pool12(nat.ptr(gpool), nat.ptr(gx), B * C, 1, st), "tron_pool12")
            if ctx.needs_input_grad[1]:
                gdense = fused.gemm_f16x3(gp, pooled, a_transposed=True, b_transposed=True, a_scale=sc)   # gp^T [Co*9, B] x pooled [B, Ci*36]
                if gdense is None:
                    gdense = gp.t() @ pooled
                gw = torch.empty(Co, C, 7, 7, dtype=torch.float32, device=pre.device)
                nat.check(L.tron_conv7_dense(nat.ptr(gdense), nat.ptr(gw), Co, C, 1, st), "tron_conv7_dense")
            if ctx.needs_input_grad[2]:
                gb = gp.view(B, Co, 9).sum((0, 2))
        return gx, gw, gb


class _PoolConv7CL(torch.autograd.Function):
    """mish(conv7(avg_pool(x))) flattened, for 26x26 / 34x34 planes (24x24 / 32x32 boards; Net/DQNNet.py:52-55): csrc/tron_head.hip's
    tron_pool_conv7_fwd / _bwd — the pooled planes kept channels-last and split, conv7 forward as the implicit GEMM the
    gradient-free head uses, its input gradient as four implicit GEMMs (one per parity class of the pooled pixel), its weight
    gradient from the two channels-last operands read transposed out of LDS.  Replaces MIOpen's NHWC igemm fwd / bwd / wrw
    kernels and their layout transposes (2.9 ms of the 21.8 ms learn step at 4 096 x 26x26)."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        from tron import _native as nat
        L = nat.lib()
        B, side = x.shape[0], x.shape[-1]
        o = (side // 2 + 1) // 2
        dev = x.device
        with torch.cuda.device(dev):
            saved = torch.empty(int(L.tron_pool_conv7_saved_bytes(B, side)), dtype=torch.uint8, device=dev)
            ws = torch.empty(int(L.tron_pool_conv7_workspace(B, side)), dtype=torch.uint8, device=dev)
            pre = torch.empty(B, o * o, 64, dtype=torch.float32, device=dev)
            y = torch.empty(B, 64 * o * o, dtype=torch.float32, device=dev)
            nat.check(L.tron_pool_conv7_fwd(nat.ptr(x), B, side, nat.ptr(weight), nat.ptr(bias), nat.ptr(saved), nat.ptr(pre), nat.ptr(y),
                                            nat.ptr(ws), nat.stream_ptr()), "tron_pool_conv7_fwd")
        ctx.save_for_backward(saved, pre, weight)
        ctx.geometry = (B, side)
        ctx.grad_scope = _current_scope()
        return y

    @staticmethod
    def backward(ctx, grad_out):
        from tron import _native as nat
        L = nat.lib()
        saved, pre, weight = ctx.saved_tensors
        B, side = ctx.geometry
        g = grad_out.contiguous()
        if not _aligned16(g):
            g = g.clone(memory_format=torch.contiguous_format)
        dev = pre.device
        gx = gw = gb = None
        with torch.cuda.device(dev):
            if ctx.needs_input_grad[0]:
                gx = torch.empty(B, 64, side, side, dtype=torch.float32, device=dev)
            if ctx.needs_input_grad[1] and not _skips_weight_gradient(ctx):
                gw = torch.empty_like(weight)
            if ctx.needs_input_grad[2]:
                gb = torch.empty(64, dtype=torch.float32, device=dev)
            ws = torch.empty(int(L.tron_pool_conv7_workspace(B, side)), dtype=torch.uint8, device=dev)
            nat.check(L.tron_pool_conv7_bwd(nat.ptr(g), nat.ptr(pre), nat.ptr(saved), nat.ptr(weight), B, side, nat.ptr(gx), nat.ptr(gw),
                                            nat.ptr(gb), nat.ptr(ws), nat.stream_ptr()), "tron_pool_conv7_bwd")
        return gx, gw, gb


class _Conv7HIP(torch.autograd.Function):
    """conv7(x) (+ bias) alone — 64 -> 64, 7x7 / stride 2 / pad 3 on 13x13 or 17x17 planes, no activation — on tron_conv7_fwd / _bwd
    (the kernels of _PoolConv7CL with NCHW tensors on both sides), for the module the K-FAC hooks sit on (kfac.py:156-189)."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        from tron import _native as nat
        L = nat.lib()
        B, ps = x.shape[0], x.shape[-1]
        o = (ps + 1) // 2
        dev = x.device
        with torch.cuda.device(dev):
            saved = torch.empty(int(L.tron_pool_conv7_saved_bytes(B, 2 * ps)), dtype=torch.uint8, device=dev)
            ws = torch.empty(int(L.tron_pool_conv7_workspace(B, 2 * ps)), dtype=torch.uint8, device=dev)
            y = torch.empty(B, 64, o, o, dtype=torch.float32, device=dev)
            nat.check(L.tron_conv7_fwd(nat.ptr(x), B, ps, nat.ptr(weight), nat.ptr(bias), nat.ptr(saved), nat.ptr(y), nat.ptr(ws),
                                       nat.stream_ptr()), "tron_conv7_fwd")
        ctx.save_for_backward(saved, weight)
        ctx.geometry = (B, ps)
        ctx.grad_scope = _current_scope()
        return y

    @staticmethod
    def backward(ctx, grad_out):
        from tron import _native as nat
        L = nat.lib()
        saved, weight = ctx.saved_tensors
        B, ps = ctx.geometry
        g = grad_out.contiguous()
        if not _aligned16(g):
            g = g.clone(memory_format=torch.contiguous_format)
        dev = g.device
        gx = gw = gb = None
        with torch.cuda.device(dev):
            if ctx.needs_input_grad[0]:
                gx = torch.empty(B, 64, ps, ps, dtype=torch.float32, device=dev)
            if ctx.needs_input_grad[1] and not _skips_weight_gradient(ctx):
                gw = torch.empty_like(weight)
            if ctx.needs_input_grad[2]:
                gb = torch.empty(64, dtype=torch.float32, device=dev)
            ws = torch.empty(int(L.tron_pool_conv7_workspace(B, 2 * ps)), dtype=torch.uint8, device=dev)
            nat.check(L.tron_conv7_bwd(nat.ptr(g), nat.ptr(saved), nat.ptr(weight), B, ps, nat.ptr(gx), nat.ptr(gw), nat.ptr(gb), nat.ptr(ws),
                                       nat.stream_ptr()), "tron_conv7_bwd")
        return gx, gw, gb


class Conv7(torch.nn.Conv2d):
    """nn.Conv2d(64, 64, 7, padding=3, stride=2) whose forward runs on the hand-written kernels where they cover the shape (f32 CUDA
    planes of side 13 / 17: 24x24 and 32x32 boards) and on the library otherwise.  Same parameters and state_dict keys, still an
    nn.Conv2d for KFACOptimizer's hooks and factor shapes."""

    def forward(self, x):
        if (_use_pool_conv7_cl and x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and x.shape[-1] == x.shape[-2]
                and x.shape[-1] in (13, 17) and 0 < x.shape[0] <= (1 << 20) and x.shape[1] == 64 and self.in_channels == 64
                and self.out_channels == 64 and self.kernel_size == (7, 7) and self.stride == (2, 2) and self.padding == (3, 3)
                and self.dilation == (1, 1) and self.groups == 1 and self.weight.dtype == torch.float32 and self.weight.is_contiguous()):
            from Net import fused
            if fused.default_math == "f16x3":
                x = x.contiguous()
                if _aligned16(x):
                    return _Conv7HIP.apply(x, self.weight, self.bias)
        return super().forward(x)


class _LinearHIP(torch.autograd.Function):
    """F.linear whose weight and bias gradients come from one launch pair of csrc/tron_dqn.hip (tron_linear_wgrad: the batch
    split over workgroups) — the library's kernels for these small outputs over a batch of 4 096 run on 1 to 144 workgroups,
    26-41 us each, plus a 7-19 us column reduction for the bias.  Forward and input gradient are the library GEMMs."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        return F.linear(x, weight, bias)

    @staticmethod
    def backward(ctx, gy):
        from tron import _native as nat
        x, weight = ctx.saved_tensors
        gy = gy.contiguous()
        gx = gy @ weight if ctx.needs_input_grad[0] else None
        gw = gb = None
        if ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2]):
            B, O, I = x.shape[0], weight.shape[0], weight.shape[1]
            L = nat.lib()
            from Net import fused
            if O * I >= _LIN_WGRAD_GEMM and fused.use_gemm:              # (as `_TailMLP`: large layers on the split-f16 GEMM)
                from Net.kfac import _pow2_scale
                gw = fused.gemm_f16x3(gy, x, a_transposed=True, b_transposed=True, a_scale=_pow2_scale(gy))
                if gw is not None:
                    return gx, gw, (gy.sum(0) if ctx.has_bias else None)
            gw = torch.empty_like(weight)
            gb = torch.empty(O, dtype=torch.float32, device=x.device) if ctx.has_bias else None
            ws = torch.empty(max(16, int(L.tron_linear_wgrad_workspace(B, O, I))), dtype=torch.uint8, device=x.device)
            with torch.cuda.device(x.device):
                nat.check(L.tron_linear_wgrad(nat.ptr(gy), nat.ptr(x), B, O, I, nat.ptr(gw), nat.ptr(gb), nat.ptr(ws), nat.stream_ptr()),
                          "tron_linear_wgrad")
        return gx, gw, gb


import os as _os
_use_linear_hip = _os.environ.get("TRON_LINEAR_HIP", "1") != "0"           # 0: the library's weight-gradient GEMMs (A/B measurements)
# tron_linear_wgrad is plain f32 FMAs on 64 x 64 tiles: right for the head's small layers (<= 150 K outputs), 239 us for fc1 behind
# 24x24 boards (256 x 3136 outputs over a batch of 4 096: 6.6 GFLOP) — from this many outputs on the weight gradient is tron_gemm_f16x3
_LIN_WGRAD_GEMM = int(_os.environ.get("TRON_LIN_WGRAD_GEMM", str(1 << 19)))


def linear(layer, x):
    """layer(x) for an nn.Linear, with its parameter gradients on tron_linear_wgrad where that applies (f32 CUDA, 2-D input, a
    batch worth splitting); the plain module otherwise."""
    if (_use_linear_hip and isinstance(layer, torch.nn.Linear) and x.is_cuda and x.dtype == torch.float32 and x.dim() == 2 and x.shape[0] >= 256
            and torch.is_grad_enabled() and layer.weight.requires_grad and layer.weight.dtype == torch.float32 and x.is_contiguous()
            and layer.weight.is_contiguous()):
        return _LinearHIP.apply(x, layer.weight, layer.bias)
    return layer(x)


class _TailMLP(torch.autograd.Function):
    """The DQN net's tail after conv7 — fc1, mish, dropout, fc2, mish, dropout, actor1, mish, actor2 (DQNNet.py:55-63) — as ONE
    autograd node.  The kernels are the ones the layer-by-layer graph runs (library GEMMs for outputs and input gradients,
    tron_linear_wgrad, tron_mish_fwd / _bwd, torch's fused dropout and its masked scale: same random stream); what goes is the
    autograd engine's trip through a dozen Python nodes between launches of 5-10 us each — the device waited ~150 us per learn
    step for the host there (scripts/learn_timeline.py: the only gaps of a step)."""

    @staticmethod
    def forward(ctx, x, p, training, w1, b1, w2, b2, w3, b3, w4, b4):
        from tron import _native as nat
        L = nat.lib()
        dev = x.device

        def act(y):
            a = torch.empty_like(y)
            nat.check(L.tron_mish_fwd(nat.ptr(y), nat.ptr(a), y.numel(), nat.stream_ptr()), "tron_mish_fwd")
            return a
        drop = training and p > 0.0
        with torch.cuda.device(dev):
            y1 = torch.addmm(b1, x, w1.t())
            a1 = act(y1)
            d1, m1 = torch._fused_dropout(a1, 1.0 - p) if drop else (a1, None)
            y2 = torch.addmm(b2, d1, w2.t())
            a2 = act(y2)
            d2, m2 = torch._fused_dropout(a2, 1.0 - p) if drop else (a2, None)
            y3 = torch.addmm(b3, d2, w3.t())
            a3 = act(y3)
            q = torch.addmm(b4, a3, w4.t())
        ctx.save_for_backward(x, y1, d1, y2, d2, y3, a3, w1, w2, w3, w4, *([m1, m2] if drop else []))
        ctx.drop, ctx.p = drop, p
        return q

    @staticmethod
    def backward(ctx, gq):
        from tron import _native as nat
        from Net import fused
        L = nat.lib()
        x, y1, d1, y2, d2, y3, a3, w1, w2, w3, w4, *masks = ctx.saved_tensors
        dev = x.device
        need = ctx.needs_input_grad                 # (x, p, training, w1, b1, ..., w4, b4)
        B = x.shape[0]
        gq = gq.contiguous()

        def wgrad(gy, inp, w):
            O, I = w.shape
            if O * I >= _LIN_WGRAD_GEMM and fused.use_gemm:    # fc1 behind 24x24 boards (256 x 3136): the split-f16 GEMM, the bias sum apart
                from Net.kfac import _pow2_scale
                gw = fused.gemm_f16x3(gy, inp, a_transposed=True, b_transposed=True, a_scale=_pow2_scale(gy))
                if gw is not None:
                    return gw, gy.sum(0)
            gw, gb = torch.empty_like(w), torch.empty(O, dtype=torch.float32, device=dev)
            ws = torch.empty(max(16, int(L.tron_linear_wgrad_workspace(B, O, I))), dtype=torch.uint8, device=dev)
            nat.check(L.tron_linear_wgrad(nat.ptr(gy), nat.ptr(inp), B, O, I, nat.ptr(gw), nat.ptr(gb), nat.ptr(ws), nat.stream_ptr()), "tron_linear_wgrad")
            return gw, gb

        def act_bwd(y, g):
            gy = torch.empty_like(y)
            nat.check(L.tron_mish_bwd(nat.ptr(y), nat.ptr(g), nat.ptr(gy), y.numel(), nat.stream_ptr()), "tron_mish_bwd")
            return gy
        scale = 1.0 / (1.0 - ctx.p) if ctx.drop else 1.0
        with torch.cuda.device(dev):
            gw4, gb4 = wgrad(gq, a3, w4)
            gy3 = act_bwd(y3, gq @ w4)
            gw3, gb3 = wgrad(gy3, d2, w3)
            g2 = gy3 @ w3
            gy2 = act_bwd(y2, torch._masked_scale(g2, masks[1], scale) if ctx.drop else g2)
            gw2, gb2 = wgrad(gy2, d1, w2)
            g1 = gy2 @ w2
            gy1 = act_bwd(y1, torch._masked_scale(g1, masks[0], scale) if ctx.drop else g1)
            gw1, gb1 = wgrad(gy1, x, w1)
            gx = gy1 @ w1 if need[0] else None
        return gx, None, None, gw1, gb1, gw2, gb2, gw3, gb3, gw4, gb4


def tail_mlp_supported(net, x):
    """Can `_TailMLP` run fc1 .. actor2 of a DQN `Net` on x [B, flat]?"""
    if not (_use_linear_hip and x.is_cuda and x.dtype == torch.float32 and x.dim() == 2 and x.shape[0] >= 256 and torch.is_grad_enabled()
            and x.is_contiguous() and _aligned16(x) and hasattr(torch, "_fused_dropout") and hasattr(torch, "_masked_scale")
            and isinstance(net.dropout, torch.nn.Dropout) and not net.dropout.inplace and 0.0 <= net.dropout.p < 1.0):
        return False
    for l in (net.fc1, net.fc2, net.actor1, net.actor2):
        if not (isinstance(l, torch.nn.Linear) and type(l).forward is torch.nn.Linear.forward and l.bias is not None and l.weight.requires_grad
                and l.bias.requires_grad and l.weight.dtype == torch.float32 and l.weight.is_contiguous() and l.weight.device == x.device
                and not l._forward_hooks and not l._forward_pre_hooks and not l._backward_hooks):
            return False
    return net.fc1.in_features == x.shape[1]


def tail_mlp(net, x):
    return _TailMLP.apply(x, float(net.dropout.p), bool(net.dropout.training), net.fc1.weight, net.fc1.bias, net.fc2.weight, net.fc2.bias,
                          net.actor1.weight, net.actor1.bias, net.actor2.weight, net.actor2.bias)


class _PoolS2(torch.autograd.Function):
    """AvgPool2d(3, stride 2, padding 1) (DQNNet.py:20,52) of even-sided planes, both directions on csrc/tron_head.hip's
    row kernels: torch's avg_pool2d backward takes 1.5 ms for 4 096 x 64 planes of 26x26 (one thread per INPUT element
    looping over windows); this is one thread per input row."""

    @staticmethod
    def forward(ctx, x):
        from tron import _native as nat
        x = x.contiguous()
        N, C, S, _ = x.shape
        y = torch.empty(N, C, S // 2, S // 2, dtype=torch.float32, device=x.device)
        with torch.cuda.device(x.device):
            nat.check(nat.lib().tron_pool_s2(nat.ptr(x), nat.ptr(y), N * C, S, nat.stream_ptr()), "tron_pool_s2")
        ctx.shape = (N, C, S)
        return y

    @staticmethod
    def backward(ctx, grad_y):
        from tron import _native as nat
        N, C, S = ctx.shape
        g = grad_y.contiguous()
        if not _aligned16(g):
            g = g.clone(memory_format=torch.contiguous_format)
        gx = torch.empty(N, C, S, S, dtype=torch.float32, device=g.device)
        with torch.cuda.device(g.device):
            nat.check(nat.lib().tron_pool_s2_bwd(nat.ptr(g), nat.ptr(gx), N * C, S, nat.stream_ptr()), "tron_pool_s2_bwd")
        return gx


def pool_s2_supported(pool, x):
    return (x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and x.shape[-1] == x.shape[-2] and x.shape[-1] in (12, 26, 34)
            and x.numel() > 0 and _aligned16(x) and isinstance(pool, torch.nn.AvgPool2d) and pool.kernel_size == 3
            and pool.stride == 2 and pool.padding == 1 and pool.count_include_pad and not pool.ceil_mode
            and pool.divisor_override is None)


def pool_s2(pool, x):
    """pool(x) with autograd, on the row kernels where they cover the shape."""
    return _PoolS2.apply(x) if pool_s2_supported(pool, x) else pool(x)


def pool_conv7_supported(pool, conv, x):
    return (x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and x.shape[-1] == 12 and x.shape[-2] == 12 and x.shape[0] > 0
            and _aligned16(x) and isinstance(pool, torch.nn.AvgPool2d) and pool.kernel_size == 3 and pool.stride == 2
            and pool.padding == 1 and pool.count_include_pad and not pool.ceil_mode and pool.divisor_override is None
            and isinstance(conv, torch.nn.Conv2d) and conv.kernel_size == (7, 7) and conv.stride == (2, 2)
            and conv.padding == (3, 3) and conv.dilation == (1, 1) and conv.groups == 1 and conv.bias is not None
            and conv.in_channels == x.shape[1] and conv.weight.is_contiguous() and conv.weight.dtype == torch.float32)


def pool_conv7_mish(pool, conv, x):
    """mish(conv7(pool(x))) as [B, Co*3*3] (already flattened in NCHW order) — DQNNet.py:52-55."""
    return _PoolConv7.apply(x, conv.weight, conv.bias)


_use_pool_conv7_cl = _os.environ.get("TRON_POOL_CONV7_HIP", "1") != "0"


def pool_conv7_cl_supported(pool, conv, x):
    """The same two layers at 26x26 / 34x34 planes (24x24 / 32x32 boards) on tron_pool_conv7_fwd / _bwd."""
    if not (_use_pool_conv7_cl and x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and x.shape[-1] == x.shape[-2]
            and x.shape[-1] in (26, 34) and 0 < x.shape[0] <= (1 << 20) and x.shape[1] == 64 and x.is_contiguous() and _aligned16(x)):
        return False
    from Net import fused
    return (fused.default_math == "f16x3" and isinstance(pool, torch.nn.AvgPool2d) and pool.kernel_size == 3 and pool.stride == 2
            and pool.padding == 1 and pool.count_include_pad and not pool.ceil_mode and pool.divisor_override is None
            and isinstance(conv, torch.nn.Conv2d) and conv.kernel_size == (7, 7) and conv.stride == (2, 2)
            and conv.padding == (3, 3) and conv.dilation == (1, 1) and conv.groups == 1 and conv.bias is not None
            and conv.in_channels == 64 and conv.out_channels == 64 and conv.weight.is_contiguous() and conv.weight.dtype == torch.float32)


def pool_conv7_cl_mish(pool, conv, x):
    """mish(conv7(pool(x))) as [B, 64 * O * O] (flattened in NCHW order) for 26x26 / 34x34 planes — DQNNet.py:52-55."""
    return _PoolConv7CL.apply(x, conv.weight, conv.bias)


def conv_bias_mish(conv, x, residual=None):
    """mish(conv(x) + residual): the 3x3 layers on the hand-written convolution kernels (forward, input and weight
    gradient: _ConvBiasMishHIP) where they cover the shape; otherwise the library convolution with the bias add, the
    residual add and the activation fused behind it (_BiasMish), or the plain composition."""
    if (x.is_cuda and x.dtype == torch.float32 and conv.bias is not None and isinstance(conv, torch.nn.Conv2d)):
        from Net import fused
        if (fused.supported(conv, x.shape[-1]) and x.shape[-2] == x.shape[-1] and _aligned16(x, residual)
                and (conv.in_channels in (3, 4) or (conv.in_channels % 8 == 0 and conv.in_channels in (32, 64)))):
            return _ConvBiasMishHIP.apply(x, conv.weight, conv.bias, residual, None)
        y = F.conv2d(x, conv.weight, None, conv.stride, conv.padding, conv.dilation, conv.groups)
        if (y.shape[2] * y.shape[3]) % 4 == 0 and y.numel() < 2 ** 32 and _aligned16(y, residual):
            return _BiasMish.apply(y, conv.bias, residual, None)
        y = y + conv.bias.view(1, -1, 1, 1)
        return mish(y if residual is None else y + residual)
    y = conv(x)
    return mish(y if residual is None else y + residual)
