"""K-FAC natural-gradient optimiser for the ACKTR path — the algorithm of the reference's
Net/kfac.py:99-254 (Kronecker-factored Fisher per layer, Martens & Grosse 2015; ACKTR, Wu et
al. 2017), written for PyTorch-ROCm:

* biases are split into `AddBias` layers so every factored module has exactly one parameter
  (kfac.py:13-24,80-96) — this also fixes the checkpoint key layout `convK.module.weight`,
  `convK.add_bias._bias`, which is kept;
* per module, running Kronecker factors  A = E[a a^T]  (inputs, via a forward pre-hook) and
  G = E[g g^T]  (output gradients of the sampled-Fisher loss, via a backward hook) with decay
  0.99 (kfac.py:41-76,156-189).  On the GPU the conv input patches come from ONE launch of
  csrc/tron_kfac.hip per chunk of samples — the fused `_extract_patches` the reference's TODO asks
  for (kfac.py:9-12); `F.unfold` would run one im2col kernel per sample — followed by one rocBLAS
  GEMM, with the reference's two normalisations folded into a single scalar;
* every `Tf` steps the factors are eigendecomposed with `torch.linalg.eigh` (hipSOLVER on
  ROCm; `torch.symeig`, kfac.py:220-223, no longer exists); eigenvalues <= 1e-6 are zeroed;
* the update is  v = Q_g [ (Q_g^T grad Q_a) / (d_g d_a^T + damping) ] Q_a^T,  KL-clipped by
  nu = min(1, sqrt(kl_clip / sum(v * grad * lr^2))), then applied by SGD with momentum 0.9 and
  lr * (1 - momentum) (kfac.py:202-254).
"""
import math
import warnings

import torch
import torch.nn as nn
import torch.nn.functional as F
import torch.optim as optim


# the first conv's input needs no gradient; its hook still receives the output gradient, which is all we use
warnings.filterwarnings("ignore", message="Full backward hook is firing when gradients are computed with respect to module outputs")


class AddBias(nn.Module):
    """A bias as its own layer; stored [C, 1] like the reference (kfac.py:13-24)."""

    def __init__(self, bias):
        super(AddBias, self).__init__()
        self._bias = nn.Parameter(bias.unsqueeze(1))
        self._kfac = None          # the KFACOptimizer whose hooks sit on this module (set by it; not part of the state)

    def forward(self, x):
        b = self._bias.t()
        return x + (b.view(1, -1) if x.dim() == 2 else b.view(1, -1, 1, 1))

    def fused_mish(self, x, residual=None):
        """mish(x + bias (+ residual)) in one pass each way (csrc/tron_nn.hip) instead of three modules' worth of
        elementwise kernels, with K-FAC's two statistics hooks of this module fed by hand: they would have seen x on the
        way in and the gradient at x + bias (+ residual) on the way back."""
        from Net.activations import bias_mish
        opt = self._kfac
        if opt is not None:
            opt._save_input(self, (x,))
        hook = None if opt is None else (lambda gp, m=self, o=opt: o._save_grad_output(m, None, (gp,)))
        return bias_mish(x, self._bias, residual, hook)


class SplitBias(nn.Module):
    """module (bias removed) followed by AddBias (kfac.py:80-96)."""

    def __init__(self, module):
        super(SplitBias, self).__init__()
        self.module = module
        self.add_bias = AddBias(module.bias.data)
        self.module.bias = None

    def forward(self, input, residual=None, act=False):
        """add_bias(module(input)); with act=True: mish(add_bias(module(input)) + residual) — what every conv layer of the
        nets' trunk is followed by (ACNet.py:97-116) — fused behind the (still hooked) module where the kernels cover it."""
        if act and not torch.is_grad_enabled():
            # gradient-free forwards (the rollouts' acting): nothing to hook, so bias, residual and activation ride in the
            # convolution kernel's epilogue instead of a pass of their own
            from Net import fused
            from Net.activations import Conv3x3, _aligned16
            m = self.module
            if (isinstance(m, Conv3x3) and input.is_cuda and input.dtype == torch.float32 and input.dim() == 4 and input.shape[0] > 0
                    and input.shape[-1] == input.shape[-2] and input.shape[1] == m.in_channels and fused.supported(m, input.shape[-1])
                    and m.in_channels in (3, 4, 32, 64) and _aligned16(input, residual)
                    and (residual is None or (residual.dtype == torch.float32 and residual.is_contiguous()))):
                return fused.conv3x3_raw(input.contiguous(), m.weight, self.add_bias._bias.reshape(-1), residual, act=True)
        if act:
            # the training graph: convolution, bias, residual and activation as ONE node (the convolution kernel's epilogue writes
            # activation and pre-activation; its backward takes mish' and the bias sums in one pass), with K-FAC's statistics hooks
            # of the convolution module and of the bias layer fed by hand — they would have seen `input` on the way in and the
            # gradient at the pre-activation on the way back (kfac.py:156-189).  A separate bias + activation pass behind the
            # hooked module (below) cost one more read and write of the layer's output each way.
            from Net import activations, fused
            m = self.module
            if (isinstance(m, activations.Conv3x3) and input.is_cuda and input.dtype == torch.float32 and input.dim() == 4 and input.shape[0] > 0
                    and input.shape[-1] == input.shape[-2] and input.shape[1] == m.in_channels and fused.supported(m, input.shape[-1])
                    and m.in_channels in (3, 4, 32, 64) and not (m.in_channels in (3, 4) and m.out_channels != 32)
                    and activations._aligned16(input, residual) and not m._forward_hooks
                    and (residual is None or (residual.dtype == torch.float32 and residual.is_contiguous()))
                    and activations.bias_mish_supported(*((residual, residual) if residual is not None else (input,)))):
                opt = self.add_bias._kfac
                hook = None
                if opt is not None:
                    opt._save_input(m, (input,))
                    opt._save_input(self.add_bias, (input,))              # (an AddBias's input factor takes the batch size only)
                    hook = lambda gp, m=m, b=self.add_bias, o=opt: (o._save_grad_output(m, None, (gp,)), o._save_grad_output(b, None, (gp,)))
                return activations._ConvBiasMishHIP.apply(input.contiguous(), m.weight, self.add_bias._bias, residual, hook)
        y = self.module(input)
        if act:
            from Net.activations import bias_mish_supported, mish
            if bias_mish_supported(y, residual):
                return self.add_bias.fused_mish(y, residual)
            y = self.add_bias(y)
            return mish(y if residual is None else y + residual)
        y = self.add_bias(y)
        return y if residual is None else y + residual


def split_biases(model):
    for name, child in model.named_children():
        if getattr(child, "bias", None) is not None and not isinstance(child, AddBias):
            model._modules[name] = SplitBias(child)
        else:
            split_biases(child)


def extract_patches(a, kernel_size, padding, stride):
    """[B, C, H, W] -> [B*OH*OW, C*kh*kw] input patches, one row per (sample, output position): the
    reference's `_extract_patches` (kfac.py:28-38) flattened.  On the GPU this is ONE launch of
    csrc/tron_kfac.hip (the fused kernel kfac.py:9-12 asks for); F.unfold runs one im2col per sample."""
    kh, kw = kernel_size
    if (a.is_cuda and a.dtype == torch.float32 and padding[0] == padding[1] and stride[0] == stride[1]):
        from tron import _native as nat
        a = a.contiguous()
        B, C, H, W = a.shape
        oh = (H + 2 * padding[0] - kh) // stride[0] + 1
        ow = (W + 2 * padding[1] - kw) // stride[1] + 1
        out = torch.empty(B * oh * ow, C * kh * kw, device=a.device, dtype=a.dtype)
        with torch.cuda.device(a.device):
            rc = nat.lib().tron_extract_patches(nat.ptr(a), B, C, H, W, kh, kw, padding[0], stride[0], nat.ptr(out),
                                                nat.stream_ptr())
        if rc == nat.OK:
            return out
        if rc != nat.ERR_UNSUPPORTED:
            nat.check(rc, "tron_extract_patches")
    cols = F.unfold(a, kernel_size, padding=padding, stride=stride)                      # [B, d, L]
    return cols.transpose(1, 2).reshape(-1, cols.size(1))


# The input factors on the hand-written Gram kernels (csrc/tron_kfac.hip: split-f16 matrix cores, upper triangle only) instead
# of extract_patches + an f32 library GEMM; TRON_KFAC_GRAM=0 keeps the library path (A/B measurements).
import os as _os
use_gram = _os.environ.get("TRON_KFAC_GRAM", "1") != "0"


def _gram_hip(a, module, scale, geometry=None, in_scale=None):
    """scale * P^T P for a conv layer's input a [B, C, H, W] (P = its patch matrix; geometry = (kh, kw, pad, stride),
    default the module's) or scale * a^T a for a Linear layer's [rows, d], on csrc/tron_kfac.hip; None where that does not
    apply (CPU tensors, tiny factors).  in_scale: a device scalar (a power of two) the input is multiplied by on its way
    into f16 — gradient tensors."""
    if not (use_gram and a.is_cuda and a.dtype == torch.float32 and a.numel() > 0):
        return None
    from tron import _native as nat
    L = nat.lib()
    a = a.contiguous()
    if a.data_ptr() % 16:
        a = a.clone()
    if a.dim() == 4:
        if geometry is None:
            if module.padding[0] != module.padding[1] or module.stride[0] != module.stride[1] or module.dilation != (1, 1):
                return None
            geometry = (module.kernel_size[0], module.kernel_size[1], module.padding[0], module.stride[0])
        B, C, H, W = a.shape
        kh, kw, pad, stride = geometry
        d = C * kh * kw
        nbytes = int(L.tron_kfac_patch_gram_workspace(B, C, H, W, kh, kw, pad, stride))
        if nbytes <= 0:
            return None
        gram = torch.empty(d, d, dtype=torch.float32, device=a.device)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=a.device)
        with torch.cuda.device(a.device):
            nat.check(L.tron_kfac_patch_gram(nat.ptr(a), B, C, H, W, kh, kw, pad, stride, float(scale), nat.ptr(in_scale),
                                             nat.ptr(gram), nat.ptr(ws), nat.stream_ptr()), "tron_kfac_patch_gram")
        return gram
    if a.dim() != 2 or a.shape[1] < 32 or a.shape[0] < 512:
        return None
    rows, d = a.shape
    nbytes = int(L.tron_kfac_gram_workspace(rows, d))
    if nbytes <= 0:
        return None
    gram = torch.empty(d, d, dtype=torch.float32, device=a.device)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=a.device)
    with torch.cuda.device(a.device):
        nat.check(L.tron_kfac_gram(nat.ptr(a), rows, d, float(scale), nat.ptr(in_scale), nat.ptr(gram), nat.ptr(ws), nat.stream_ptr()),
                  "tron_kfac_gram")
    return gram


def _pow2_scale(g):
    """A device scalar 2^k that brings max |g| to [2^9, 2^10) (f16's comfortable range after the kernels' 2^-6), without
    reading anything back; 1 for an all-zero tensor."""
    if g.is_cuda and g.dtype == torch.float32 and g.is_contiguous() and g.data_ptr() % 16 == 0:
        from tron import _native as nat
        out = torch.zeros(4, dtype=torch.float32, device=g.device)          # {scale, max |g|, scratch, scratch}: tron_absmax_pow2, one launch
        with torch.cuda.device(g.device):
            nat.check(nat.lib().tron_absmax_pow2(nat.ptr(g), g.numel(), 16, nat.ptr(out), nat.stream_ptr()), "tron_absmax_pow2")
        return out[:1]
    lo, hi = torch.aminmax(g)
    m = torch.maximum(hi, -lo).to(torch.float32)
    e = torch.frexp(m)[1]                                   # m = f 2^e, f in [0.5, 1)
    s = torch.ldexp(torch.ones((), dtype=torch.float32, device=g.device), (16 - e).clamp(-60, 60))     # (its square must stay finite)
    return torch.where(m > 0, s, torch.ones_like(s)).reshape(1).contiguous()


def cov_inputs(a, module, batch=None):
    """A-factor sample of one batch (kfac.py:41-58).  `batch` = size of the WHOLE batch when `a` is
    only a micro-batch of it (the result is then this micro-batch's additive share)."""
    if not torch.is_tensor(a):                      # a PX16 image (Net/fused.py): the weight-stationary trunk's layer inputs (3x3 / pad 1 / stride 1)
        n = a.shape[0]
        got = a.kfac_input_gram(1.0 / ((n if batch is None else batch) * float(a.shape[2] * a.shape[3]) ** 2))
        return got if got is not None else cov_inputs(a.float(), module, batch)
    share = 1.0 if batch is None else a.size(0) / batch
    batch = a.size(0) if batch is None else batch
    if isinstance(module, nn.Conv2d):
        oh = (a.size(2) + 2 * module.padding[0] - module.kernel_size[0]) // module.stride[0] + 1
        ow = (a.size(3) + 2 * module.padding[1] - module.kernel_size[1]) // module.stride[1] + 1
        d = a.size(1) * module.kernel_size[0] * module.kernel_size[1]
        got = _gram_hip(a, module, 1.0 / (batch * float(oh * ow) ** 2))
        if got is not None:
            return got
        if a.is_cuda:
            # rows/(oh*ow) then rows^T (rows/batch) of the reference is P^T P / (batch (oh ow)^2): the patch
            # matrix goes into the GEMM unscaled, in chunks of <= 1 GB so 16 384 envs x 5 steps stay bounded
            chunk = max(1, (256 << 20) // max(d * oh * ow, 1))
            acc = torch.zeros(d, d, device=a.device, dtype=a.dtype)
            for i in range(0, a.size(0), chunk):
                p = extract_patches(a[i:i + chunk], module.kernel_size, module.padding, module.stride)
                acc.addmm_(p.t(), p)       # (block-triangular Gram products were tried: slower on rocBLAS, 51 s vs 32 s)
            return acc.mul_(1.0 / (batch * float(oh * ow) ** 2))
        # im2col in batch chunks of <= ~256 MB
        chunk = max(1, min(a.size(0), (64 << 20) // max(d * oh * ow, 1)))
        if chunk >= a.size(0):
            rows = extract_patches(a, module.kernel_size, module.padding, module.stride) / (oh * ow)
            return rows.t() @ (rows / batch)
        acc = torch.zeros(d, d, device=a.device, dtype=a.dtype)
        for i in range(0, a.size(0), chunk):
            rows = extract_patches(a[i:i + chunk], module.kernel_size, module.padding, module.stride) / (oh * ow)
            acc.addmm_(rows.t(), rows / batch)
        return acc
    if isinstance(module, AddBias):
        return torch.full((1, 1), share, device=a.device, dtype=a.dtype)   # ones(B,1)^T ones(B,1) / B
    got = _gram_hip(a, module, 1.0 / batch)
    return got if got is not None else a.t() @ (a / batch)


def cov_grads(g, module, batch=None):
    """G-factor sample of one batch (kfac.py:61-76).  The gradients are those of a loss averaged
    over the whole batch; `batch` is its size when `g` covers only a micro-batch."""
    scale = 1.0 if batch is None else batch / g.size(0)        # rows of the whole batch / rows here
    batch = g.size(0) if batch is None else batch
    if isinstance(module, nn.Conv2d):
        oh, ow = g.size(2), g.size(3)
        if g.is_cuda and g.dtype == torch.float32 and use_gram and g.size(1) >= 16:
            # g_ = g (oh ow) batch as rows (sample, position); g_^T g_ / (rows scale): the NCHW tensor is its own 1x1 patch matrix
            rows = g.size(0) * oh * ow
            got = _gram_hip(g, module, (float(oh * ow) * batch) ** 2 / (rows * scale), geometry=(1, 1, 0, 1), in_scale=_pow2_scale(g))
            if got is not None:
                return got
        g = g.permute(0, 2, 3, 1).reshape(-1, g.size(1)) * (oh * ow)
    elif isinstance(module, AddBias):
        g = g.reshape(g.size(0), g.size(1), -1).sum(-1)
    if g.is_cuda and g.dtype == torch.float32 and g.dim() == 2:
        got = _gram_hip(g, module, float(batch) ** 2 / (g.size(0) * scale), in_scale=_pow2_scale(g))
        if got is not None:
            return got
    g_ = g * batch
    return g_.t() @ (g_ / (g.size(0) * scale))


class KFACOptimizer(optim.Optimizer):
    def __init__(self, model, lr=0.25, momentum=0.9, stat_decay=0.99, kl_clip=0.001, damping=1e-2, weight_decay=0,
                 fast_cnn=False, Ts=1, Tf=10):
        if fast_cnn:
            raise NotImplementedError("fast_cnn factors are not used by the reference's trainer")
        split_biases(model)
        super(KFACOptimizer, self).__init__(model.parameters(), dict())
        self.model = model
        self.modules = [m for m in model.modules() if isinstance(m, (nn.Linear, nn.Conv2d, AddBias))]
        for m in self.modules:
            assert len(list(m.parameters(recurse=False))) == 1, "one parameter per factored module"
            m.register_forward_pre_hook(self._save_input)
            m.register_full_backward_hook(self._save_grad_output)
            if isinstance(m, AddBias):
                m._kfac = self                    # (AddBias.fused_mish calls the two hooks itself)
        self.steps = 0
        self.acc_stats = False
        self._whole_batch = None          # set by accumulate(): hooks then add micro-batch shares
        self._sum_aa, self._sum_gg = {}, {}
        self.m_aa, self.m_gg = {}, {}
        self.Q_a, self.Q_g, self.d_a, self.d_g = {}, {}, {}, {}
        self.momentum, self.stat_decay, self.lr = momentum, stat_decay, lr
        self.kl_clip, self.damping, self.weight_decay = kl_clip, damping, weight_decay
        self.Ts, self.Tf = Ts, Tf
        self.optim = optim.SGD(model.parameters(), lr=self.lr * (1 - self.momentum), momentum=self.momentum)

    # ---- factor statistics -----------------------------------------------------------------
    @staticmethod
    def _mean_over_ranks(sample):
        """One process per GPU (torch.distributed initialised, world > 1): a factor sample is the mean of the ranks' samples — each
        rank's batch is its share of one global batch (A = E[a a^T] and G = E[g g^T] over that batch, the gradients being those of
        per-sample losses: cov_grads), every rank the same size.  A single process: the sample as it is."""
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            from DDQN import all_reduce_mean_
            all_reduce_mean_(sample)
        return sample

    def _running(self, store, module, sample):
        if self.steps == 0:
            store[module] = sample.clone()
        store[module].mul_(self.stat_decay).add_(sample, alpha=1 - self.stat_decay)      # kfac.py:79-83

    def _save_input(self, module, inputs):
        if torch.is_grad_enabled() and self.steps % self.Ts == 0:
            with torch.no_grad():
                a = inputs[0].detach() if torch.is_tensor(inputs[0]) else inputs[0]
                if self._whole_batch is None:
                    self._running(self.m_aa, module, self._mean_over_ranks(cov_inputs(a, module)))
                else:
                    part = cov_inputs(a, module, self._whole_batch)
                    self._sum_aa[module] = part if module not in self._sum_aa else self._sum_aa[module].add_(part)

    def _save_grad_output(self, module, grad_input, grad_output):
        if self.acc_stats:
            with torch.no_grad():
                if self._whole_batch is None:
                    self._running(self.m_gg, module, self._mean_over_ranks(cov_grads(grad_output[0].detach(), module)))
                else:
                    part = cov_grads(grad_output[0].detach(), module, self._whole_batch)
                    self._sum_gg[module] = part if module not in self._sum_gg else self._sum_gg[module].add_(part)

    # ---- micro-batched statistics: one running-average update per step, like a single big batch ----
    def begin_accumulate(self, whole_batch):
        self._whole_batch = int(whole_batch)
        self._sum_aa, self._sum_gg = {}, {}

    def end_accumulate(self):
        for module in self.modules:                 # (a fixed order: with one rank per GPU every sample is a collective)
            if module in self._sum_aa:
                self._running(self.m_aa, module, self._mean_over_ranks(self._sum_aa[module]))
        for module in self.modules:
            if module in self._sum_gg:
                self._running(self.m_gg, module, self._mean_over_ranks(self._sum_gg[module]))
        self._whole_batch = None
        self._sum_aa, self._sum_gg = {}, {}

    # ---- the step -------------------------------------------------------------------------------
    @torch.no_grad()
    def step(self):
        if self.weight_decay > 0:
            for p in self.model.parameters():
                p.grad.add_(p, alpha=self.weight_decay)
        la = self.damping + self.weight_decay
        updates = {}
        for m in self.modules:
            p = next(m.parameters(recurse=False))
            if self.steps % self.Tf == 0:
                self.d_a[m], self.Q_a[m] = torch.linalg.eigh(self.m_aa[m])
                self.d_g[m], self.Q_g[m] = torch.linalg.eigh(self.m_gg[m])
                self.d_a[m].mul_((self.d_a[m] > 1e-6).float())
                self.d_g[m].mul_((self.d_g[m] > 1e-6).float())
            grad = p.grad.reshape(p.grad.size(0), -1) if isinstance(m, nn.Conv2d) else p.grad
            v1 = self.Q_g[m].t() @ grad @ self.Q_a[m]
            v2 = v1 / (self.d_g[m].unsqueeze(1) * self.d_a[m].unsqueeze(0) + la)
            updates[p] = (self.Q_g[m] @ v2 @ self.Q_a[m].t()).view_as(p.grad)
        vg_sum = 0.0
        for p in self.model.parameters():
            vg_sum = vg_sum + (updates[p] * p.grad * self.lr * self.lr).sum()
        nu = min(1.0, math.sqrt(self.kl_clip / float(vg_sum)))
        for p in self.model.parameters():
            p.grad.copy_(updates[p]).mul_(nu)
        self.optim.step()
        self.steps += 1
