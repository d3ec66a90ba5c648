"""Actor-critic networks of the reference's ACKTR path (Net/ACNet.py:6-397) on PyTorch-ROCm.

Same class names, attribute names and parameter shapes (so `.bak` state_dicts interchange, also
after KFACOptimizer has split the biases), same forward maths.  Differences, all deliberate:
the shared 7-conv trunk is written once; the board side is a parameter (fc1 is sized from it,
DQNNet.conv7_side) instead of the hard-wired 12x12; nothing calls `.cuda()` — tensors follow the
module's device (ACNet.py:94,164,233,300 pin the reference to a GPU).

  MapNet  (ACNet.py:335-397)  4 input planes (pop_up + prob_map), no env vector
  TestNet (ACNet.py:59-126)   3 planes, env scalar concatenated before the heads (129 wide)
  Net3    (ACNet.py:128-196)  3 planes, fc1 output gated by tanh(fc_env(scalar))
  Net4    (ACNet.py:198-263)  3 planes, env scalar concatenated before fc2 (257 wide)
  Mulnet  (ACNet.py:265-333)  3 planes, fc1 output gated by tanh(fc_env([degree, weight]))
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from config import MAP_WIDTH
from Net.DQNNet import conv7_side
from Net.activations import (mish as _mish, Conv3x3 as _Conv3x3, Conv7 as _Conv7, pool_s2 as _pool_s2, conv_bias_mish as _conv_bias_mish,
                             pool_conv7_cl_mish as _pool_conv7_cl_mish, pool_conv7_cl_supported as _pool_conv7_cl_supported)
from Net.kfac import SplitBias as _SplitBias


class Net(nn.Module):
    """Policy/value interface shared by all nets (ACNet.py:6-57)."""

    wants_prob_plane = False        # MapNet: Game.main_loop feeds pop_up + prob_map (game.py:297)

    def act(self, x, env_prob=None):
        """Sample an action from softmax(actor logits) (ACNet.py:14-26)."""
        value, actor_output = self(x, env_prob) if env_prob is not None else self(x)
        return F.softmax(actor_output, dim=1).multinomial(num_samples=1)

    def deterministic_act(self, x, env_prob=None):          # ACNet.py:28-31
        value, actor_output = self(x, env_prob) if env_prob is not None else self(x)
        return torch.argmax(actor_output, dim=1)

    def get_value(self, x, env_prob=None):                  # ACNet.py:33-39
        value, actor_output = self(x, env_prob) if env_prob is not None else self(x)
        return value

    def evaluate_actions(self, x, actions, env_prob=None):
        """(value, log pi(a|s), mean entropy) for a batch (ACNet.py:41-54)."""
        value, actor_output = self(x, env_prob) if env_prob is not None else self(x)
        log_probs = F.log_softmax(actor_output, dim=1)
        action_log_probs = log_probs.gather(1, actions.detach())
        probs = F.softmax(actor_output, dim=1)
        entropy = -(log_probs * probs).sum(-1).mean()
        return value, action_log_probs, entropy

    @staticmethod
    def mish(x):                                            # ACNet.py:56-57, fused (Net/activations.py)
        return _mish(x)

    # ---- the trunk every net shares (e.g. ACNet.py:97-116) -------------------------------
    def _build_trunk(self, in_channels, width):
        self.side = width + 2
        # Conv3x3: nn.Conv2d with its forward / input gradient / weight gradient on csrc/tron_conv_f16.hip, tron_conv_wgrad*.hip
        # where they cover the shape (boards 10x10, 24x24, 32x32) — still the module K-FAC hooks (Net/kfac.py)
        self.conv1 = _Conv3x3(in_channels, 32, 3, padding=1)
        self.conv2 = _Conv3x3(32, 32, 3, padding=1)
        self.conv3 = _Conv3x3(32, 32, 3, padding=1)
        self.conv4 = _Conv3x3(32, 64, 3, padding=1)
        self.conv5 = _Conv3x3(64, 64, 3, padding=1)
        self.conv6 = _Conv3x3(64, 64, 3, padding=1)
        self.pool = nn.AvgPool2d(kernel_size=3, padding=1, stride=2)
        self.conv7 = _Conv7(64, 64, 7, padding=3, stride=2)      # (an nn.Conv2d; tron_conv7_fwd / _bwd at 13x13 / 17x17 pooled planes)
        self.flat = 64 * conv7_side(self.side) ** 2
        self.fc1 = nn.Linear(self.flat, 256)

    def _conv_act(self, conv, x, residual=None):
        """activation(conv(x) + residual).  With the nets' own mish: a plain conv layer goes through `conv_bias_mish`
        (convolution, bias, residual and activation in one kernel where the HIP kernels cover the shape); after
        KFACOptimizer split the bias off (Net/kfac.py::SplitBias) the hooked convolution module runs first and bias +
        residual + activation are one pass behind it."""
        if self.activation is Net.mish or self.activation is self.mish:
            if isinstance(conv, _SplitBias):
                return conv(x, residual=residual, act=True)
            if isinstance(conv, nn.Conv2d) and conv.bias is not None:
                return _conv_bias_mish(conv, x, residual)
        y = conv(x)
        return self.activation(y if residual is None else y + residual)

    def _trunk(self, x):
        """conv1..conv7 + fc1 with dropout; returns the [B, 256] feature."""
        a = self.activation
        fused_trunk = self._trunk_px(x) if (a is Net.mish or a is self.mish) else None
        if fused_trunk is not None:
            x = fused_trunk
        else:
            x = self._conv_act(self.conv1, x)
            idx = x
            x = self._conv_act(self.conv2, x)
            x = self._conv_act(self.conv3, x, idx)
            x = self._conv_act(self.conv4, x)
            idx = x
            x = self._conv_act(self.conv5, x)
            x = self._conv_act(self.conv6, x, idx)
        if ((a is Net.mish or a is self.mish) and isinstance(self.conv7, nn.Conv2d) and torch.is_grad_enabled()
                and _pool_conv7_cl_supported(self.pool, self.conv7, x)):
            x = _pool_conv7_cl_mish(self.pool, self.conv7, x)   # 24x24 / 32x32 boards, plain A2C: tron_pool_conv7_fwd / _bwd
        else:                                 # (under K-FAC conv7 is the hooked SplitBias module: its factors need the pooled planes)
            x = _pool_s2(self.pool, x)        # (csrc/tron_head.hip's row kernels at 12 / 26 / 34, both directions; else self.pool)
            x = self._conv_act(self.conv7, x)
            x = x.reshape(-1, self.flat)
        return self.dropout(a(self.fc1(x)))

    def _trunk_px(self, x):
        """conv1 .. conv6 as one autograd node on the weight-stationary kernels (Net/activations.py::_ACTrunkPX) where they cover
        the shapes — with K-FAC's hooks of the six layers fed by hand when the optimizer has split the biases; None otherwise."""
        from Net import activations
        convs = [self.conv1, self.conv2, self.conv3, self.conv4, self.conv5, self.conv6]
        split = [isinstance(c, _SplitBias) for c in convs]
        if all(split):
            mods = [c.module for c in convs]
            if not all(isinstance(m, _Conv3x3) and not m._forward_hooks for m in mods):
                return None
            wb = [t for c in convs for t in (c.module.weight, c.add_bias._bias)]
            hooks = activations.TrunkHooks(convs)
        elif not any(split) and all(isinstance(c, nn.Conv2d) and c.bias is not None and not c._forward_hooks and not c._forward_pre_hooks
                                    and not c._backward_hooks for c in convs):
            wb = [t for c in convs for t in (c.weight, c.bias)]
            hooks = None
        else:
            return None
        if not torch.is_grad_enabled():                                  # the rollouts' acting: nothing to hook
            if not activations.ac_trunk_px_supported(x, wb[0::2], need_grad=False):
                return None
            return activations.ac_trunk_infer(x.contiguous(), wb[0::2], wb[1::2])
        if not activations.ac_trunk_px_supported(x, wb[0::2]):
            return None
        return activations._ACTrunkPX.apply(x.contiguous(), hooks, *wb)

    def _heads(self, x):
        a = self.activation
        actor_output = self.actor2(a(self.actor1(x)))
        critic_output = self.critic3(a(self.critic2(a(self.critic1(x)))))
        return critic_output, actor_output

    def _device(self):
        return self.conv1.module.weight.device if hasattr(self.conv1, "module") else self.conv1.weight.device


def _finish(net):
    net.dropout = nn.Dropout(p=0.2)
    net.activation = net.mish


class TestNet(Net):
    def __init__(self, width=MAP_WIDTH):
        super(TestNet, self).__init__()
        self._build_trunk(3, width)
        self.fc2 = nn.Linear(256, 128)
        self.actor1 = nn.Linear(129, 64)
        self.actor2 = nn.Linear(64, 4)
        self.critic1 = nn.Linear(129, 64)
        self.critic2 = nn.Linear(64, 16)
        self.critic3 = nn.Linear(16, 1)
        _finish(self)

    def forward(self, x, env_prob):
        dev = self._device()
        env_prob = env_prob.unsqueeze(1).detach().to(dev)
        x = self._trunk(x.to(dev))
        x = self.dropout(self.activation(self.fc2(x)))
        return self._heads(torch.cat([x, env_prob], 1))


class Net3(Net):
    def __init__(self, width=MAP_WIDTH):
        super(Net3, self).__init__()
        self._build_trunk(3, width)
        self.fc_env = nn.Linear(1, 256)
        self.fc2 = nn.Linear(256, 128)
        self.actor1 = nn.Linear(128, 32)
        self.actor2 = nn.Linear(32, 4)
        self.critic1 = nn.Linear(128, 32)
        self.critic2 = nn.Linear(32, 8)
        self.critic3 = nn.Linear(8, 1)
        _finish(self)

    def forward(self, x, env_prob):
        dev = self._device()
        env_prob = env_prob.unsqueeze(1).to(dev)
        x = self._trunk(x.to(dev))
        x = x.mul(torch.tanh(self.fc_env(env_prob)))
        x = self.dropout(self.activation(self.fc2(x)))
        return self._heads(x)


class Net4(Net):
    def __init__(self, width=MAP_WIDTH):
        super(Net4, self).__init__()
        self._build_trunk(3, width)
        self.fc2 = nn.Linear(257, 128)
        self.actor1 = nn.Linear(128, 64)
        self.actor2 = nn.Linear(64, 4)
        self.critic1 = nn.Linear(128, 64)
        self.critic2 = nn.Linear(64, 16)
        self.critic3 = nn.Linear(16, 1)
        _finish(self)

    def forward(self, x, env_prob):
        dev = self._device()
        env_prob = env_prob.unsqueeze(1).detach().to(dev)
        x = self._trunk(x.to(dev))
        x = self.dropout(self.activation(self.fc2(torch.cat([x, env_prob], 1))))
        return self._heads(x)


class Mulnet(Net):
    def __init__(self, width=MAP_WIDTH):
        super(Mulnet, self).__init__()
        self._build_trunk(3, width)
        self.fc_env = nn.Linear(2, 256)
        self.fc2 = nn.Linear(256, 128)
        self.actor1 = nn.Linear(128, 32)
        self.actor2 = nn.Linear(32, 4)
        self.critic1 = nn.Linear(128, 32)
        self.critic2 = nn.Linear(32, 8)
        self.critic3 = nn.Linear(8, 1)
        _finish(self)

    def forward(self, x, env_prob):
        dev = self._device()
        env_prob = env_prob.to(dev)
        x = self._trunk(x.to(dev))
        x = x.mul(torch.tanh(self.fc_env(env_prob)))
        x = self.dropout(self.activation(self.fc2(x)))
        return self._heads(x)


class MapNet(Net):
    wants_prob_plane = True

    def __init__(self, width=MAP_WIDTH):
        super(MapNet, self).__init__()
        self._build_trunk(4, width)
        self.fc2 = nn.Linear(256, 128)
        self.actor1 = nn.Linear(128, 32)
        self.actor2 = nn.Linear(32, 4)
        self.critic1 = nn.Linear(128, 32)
        self.critic2 = nn.Linear(32, 8)
        self.critic3 = nn.Linear(8, 1)
        _finish(self)

    def forward(self, x):
        x = self._trunk(x.to(self._device()))
        x = self.dropout(self.activation(self.fc2(x)))
        return self._heads(x)
