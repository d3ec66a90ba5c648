"""The reference's DQN network (Net/DQNNet.py:6-66) on PyTorch-ROCm, with its two defects
fixed: `mish` is defined here, and the input channel count and the
board side are parameters, so fc1 is sized from the grid instead of hard-wired to 64*3*3.
Parameter names/shapes are the reference's (22 tensors, 501 924 parameters at 4 channels,
12x12), so `torch.save(state_dict)` .bak files interchange.

The network is compute-bound on matrix throughput (36 MFLOP per 12x12 sample), so its layers run on the
hand-written kernels of csrc/: the 3x3 convolutions forward / input gradient / weight gradient (tron_conv*.hip), the
pooling + conv7 + linear + arg-max head of gradient-free forwards (tron_head.hip, `Net.infer`), pooling + conv7 as
GEMMs on conv7's dense form when training.  Shapes those kernels do not cover, and CPU tensors, take the PyTorch
modules."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from config import MAP_WIDTH
from Net.activations import (mish as _mish, conv_bias_mish as _conv_bias_mish, pool_conv7_mish as _pool_conv7_mish,
                             pool_conv7_supported as _pool_conv7_supported, pool_s2 as _pool_s2, linear as _linear,
                             pool_conv7_cl_mish as _pool_conv7_cl_mish, pool_conv7_cl_supported as _pool_conv7_cl_supported)


def _pool_is_reference(pool):
    """AvgPool2d(kernel_size=3, padding=1, stride=2) as DQNNet.py:20 builds it."""
    return (isinstance(pool, nn.AvgPool2d) and pool.kernel_size == 3 and pool.stride == 2 and pool.padding == 1
            and pool.count_include_pad and not pool.ceil_mode and pool.divisor_override is None)


def conv7_side(side):
    """Spatial side after avgpool(k3,s2,p1) then conv7(k7,s2,p3) (DQNNet.py:20,22)."""
    pooled = (side + 2 - 3) // 2 + 1
    return (pooled + 6 - 7) // 2 + 1


class Net(nn.Module):
    def __init__(self, in_channels=4, width=MAP_WIDTH):
        super(Net, self).__init__()
        self.in_channels = in_channels
        self.side = width + 2
        self.conv1 = nn.Conv2d(in_channels, 32, 3, padding=1)        # DQNNet.py:10
        self.conv2 = nn.Conv2d(32, 32, 3, padding=1)
        self.conv3 = nn.Conv2d(32, 32, 3, padding=1)
        self.conv4 = nn.Conv2d(32, 64, 3, padding=1)
        self.conv5 = nn.Conv2d(64, 64, 3, padding=1)
        self.conv6 = nn.Conv2d(64, 64, 3, padding=1)
        self.pool = nn.AvgPool2d(kernel_size=3, padding=1, stride=2)
        self.conv7 = nn.Conv2d(64, 64, 7, padding=3, stride=2)
        self.flat = 64 * conv7_side(self.side) ** 2                   # 576 at 12x12 (DQNNet.py:24,55)
        self.fc1 = nn.Linear(self.flat, 256)
        self.fc2 = nn.Linear(256, 128)
        self.actor1 = nn.Linear(128, 64)
        self.actor2 = nn.Linear(64, 4)
        self.dropout = nn.Dropout(p=0.2)
        self.activation = self.mish
        self.fuse_trunk = True       # conv1..conv6 as one autograd node where the HIP kernels cover the shape (Net/activations.py)

    @staticmethod
    def mish(x):
        """x * tanh(softplus(x)) (ACNet.py:56-57) as ONE kernel forward and one backward (Net/activations.py);
        the composed form is three elementwise launches per activation."""
        return _mish(x)

    def forward_codes(self, codes, plane4=0.0):
        """forward() from the env's int8 observation codes [B, S, S] (Map.state_for_player, map.py:67-84) instead of
        util.pop_up's f32 planes: conv1 builds the planes while it stages its input (the learner's batch comes out of
        the replay ring as codes: tron_replay_sample_codes).  Where the HIP kernels do not cover the shape the planes
        are built explicitly."""
        from Net import fused
        from Net.activations import conv1_codes_mish, trunk_mish, trunk_supported, body_mish, body_px_supported
        side = codes.shape[-1]
        if codes.is_cuda and fused.supported(self.conv1, side) and (self.activation is Net.mish or self.activation is self.mish):
            codes = codes.reshape(-1, side, side)
            if self.fuse_trunk and torch.is_grad_enabled() and body_px_supported(self, codes):
                return self._after_conv7(body_mish(self, codes, plane4))          # conv1 .. conv7 as one node (Net/activations.py::_BodyPX)
            if self.fuse_trunk and trunk_supported(self, codes):
                return self._after_trunk(trunk_mish(self, codes, plane4))
            return self._after_conv1(conv1_codes_mish(self.conv1, codes, plane4))
        from tron.vec import pop_up_planes
        x = pop_up_planes(codes.reshape(-1, side, side))
        if self.in_channels == 4:
            x = torch.cat([x, torch.full_like(x[:, :1], plane4)], 1)
        return self(x)

    def forward(self, x):                         # DQNNet.py:33-63
        x = x.to(self.conv1.weight.device)
        if self.activation is not Net.mish and self.activation is not self.mish:   # a caller swapped the activation
            return self._forward_plain(x)
        if self.fuse_trunk and torch.is_grad_enabled():
            from Net.activations import trunk_mish, trunk_supported
            if trunk_supported(self, x):
                return self._after_trunk(trunk_mish(self, x))
        return self._after_conv1(_conv_bias_mish(self.conv1, x))

    def _after_conv1(self, x):
        idx = x
        x = _conv_bias_mish(self.conv2, x)
        x = _conv_bias_mish(self.conv3, x, idx)
        x = _conv_bias_mish(self.conv4, x)
        idx = x
        x = _conv_bias_mish(self.conv5, x)
        x = _conv_bias_mish(self.conv6, x, idx)
        return self._after_trunk(x)

    def _after_trunk(self, x):
        if _pool_conv7_supported(self.pool, self.conv7, x):
            x = _pool_conv7_mish(self.pool, self.conv7, x)              # the two layers as GEMMs on conv7's dense form
        elif _pool_conv7_cl_supported(self.pool, self.conv7, x):
            x = _pool_conv7_cl_mish(self.pool, self.conv7, x)           # 24x24 / 32x32 boards: implicit GEMMs, both directions
        else:
            x = _pool_s2(self.pool, x)                                   # (24x24 boards: the row kernels, both directions)
            x = _conv_bias_mish(self.conv7, x)
            x = x.reshape(-1, self.flat)
        return self._after_conv7(x)

    def _after_conv7(self, x):
        if self.fuse_trunk and (self.activation is Net.mish or self.activation is self.mish):
            from Net.activations import tail_mlp, tail_mlp_supported
            if tail_mlp_supported(self, x):
                return tail_mlp(self, x)                                  # fc1 .. actor2 as one autograd node
        x = self.dropout(self.activation(_linear(self.fc1, x)))
        x = self.dropout(self.activation(_linear(self.fc2, x)))
        return _linear(self.actor2, self.activation(_linear(self.actor1, x)))

    def _forward_plain(self, x):
        x = self.activation(self.conv1(x))
        idx = x
        x = self.activation(self.conv2(x))
        x = self.activation(self.conv3(x) + idx)
        x = self.activation(self.conv4(x))
        idx = x
        x = self.activation(self.conv5(x))
        x = self.activation(self.conv6(x) + idx)
        x = self.pool(x)
        x = self.activation(self.conv7(x))
        x = x.reshape(-1, self.flat)
        x = self.dropout(self.activation(self.fc1(x)))
        x = self.dropout(self.activation(self.fc2(x)))
        return self.actor2(self.activation(self.actor1(x)))

    def act(self, x, env_prob=None):              # DQNNet.py:64-66 (env_prob accepted for Game.main_loop)
        if not self.training and torch.is_tensor(x) and x.is_cuda and x.dtype == torch.float32 and x.dim() == 4:
            return self.infer(x, greedy=True).long()      # eval mode: the gradient-free kernels, arg-max taken in the head
        return torch.argmax(self(x), dim=1)

    def infer_path(self, x, codes=False):
        """Which implementation `infer` runs for this input: "ws-chain" (csrc/tron_conv_ws.hip + tron_head.hip), "layer-kernels"
        (csrc/tron_conv_f16.hip / tron_conv.hip, head on tron_head.hip or the libraries) or "module" (PyTorch-ROCm)."""
        from Net import fused
        side = x.shape[-1]
        if not (x.is_cuda and fused.supported(self.conv1, side) and fused.supported(self.conv6, side)
                and (codes or x.dtype == torch.float32)):
            return "module"
        if codes and fused.default_math == "f16x3" and fused.ws_supported(self, side) and fused.head_supported(self, side):
            return "ws-chain"
        return "layer-kernels"

    def infer(self, x, codes=False, plane4=0.0, greedy=False):
        """Q-values without autograd and without dropout (what `eval()` + `no_grad()` give, DDQN.py:90-110,129-142)
        on the hand-written HIP path.  From the env's int8 observation codes (codes=True) at 12x12 and 26x26: the
        weight-stationary chain of csrc/tron_conv_ws.hip (conv1 as a table sum over the codes, conv2..conv6 on the split-f16
        matrix cores with the weights in registers and the activations as PX16 images) into csrc/tron_head.hip (pooling,
        conv7, the four linear layers, arg-max).  From f32 planes: csrc/tron_conv_f16.hip / tron_conv.hip layer by layer, then
        the same head.  Falls back to the module's own forward for shapes the kernels do not cover (other sides, CPU
        tensors) — `infer_path(x, codes)` says which of the three a call takes.  greedy=True returns the arg-max action per row as int8 instead (`Net.act`, DQNNet.py:64-66), taken
        inside the head kernel where that runs."""
        from Net import fused
        side = x.shape[-1]
        # what the kernels assume about the input is checked here: they are handed raw pointers
        if codes:
            if x.dtype != torch.int8 or x.dim() < 2 or x.shape[-2] != side:
                raise TypeError(f"infer(codes=True) takes int8 observation codes [..., S, S], got {tuple(x.shape)} {x.dtype}")
        elif x.dim() != 4 or x.shape[-2] != side or x.shape[1] != self.conv1.in_channels:
            raise TypeError(f"infer takes planes [B, {self.conv1.in_channels}, S, S], got {tuple(x.shape)}")
        with torch.no_grad():
            if not (x.is_cuda and fused.supported(self.conv1, side) and fused.supported(self.conv6, side)
                    and (codes or x.dtype == torch.float32)):
                if codes:
                    from tron.vec import pop_up_planes
                    x = pop_up_planes(x.reshape(-1, side, side))
                    if self.in_channels == 4:
                        x = torch.cat([x, torch.full_like(x[:, :1], plane4)], 1)
                was_training = self.training
                self.eval()
                try:
                    return self(x).argmax(1).to(torch.int8) if greedy else self(x)
                finally:
                    self.train(was_training)
            if codes and fused.default_math == "f16x3" and fused.ws_supported(self, side) and fused.head_supported(self, side):
                # the weight-stationary chain: PX16 images from conv1's output to the head's pooling, no f32 activation tensor
                x = fused.trunk_px(self, x.reshape(-1, side, side), plane4, want="head")
                return fused.head(self, x, want_q=False, want_greedy=True)[1] if greedy else fused.head(self, x)
            x = fused.trunk(self, x.reshape(-1, side, side) if codes else x, codes=codes, plane4=plane4)
            if fused.default_math == "f16x3" and fused.head_supported(self, side):
                return fused.head(self, x, want_q=False, want_greedy=True)[1] if greedy else fused.head(self, x)
            if side in (12, 26) and _pool_is_reference(self.pool):
                x = fused.pool_s2(x)                                     # (torch's avg_pool2d runs at a quarter of the memory rate)
            else:
                x = self.pool(x)
            x = _conv_bias_mish(self.conv7, x)
            x = x.reshape(-1, self.flat)
            x = _mish(self.fc1(x))
            x = _mish(self.fc2(x))
            q = self.actor2(_mish(self.actor1(x)))
            return q.argmax(1).to(torch.int8) if greedy else q
