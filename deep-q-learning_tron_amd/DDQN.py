"""Double-DQN learner — the reference's DDQN.py (Agent :34-165, ReplayBuffer :167-203,
train :206-346) with the replay memory resident in HBM and a batched trainer on VecTron.

Two ways in:
  * drop-in: `Agent()`, `.action(obs)`, `.step(s, a, r, s2, done)`, `.learn`, `.soft_update`,
    `ReplayBuffer(action_size, buffer_size, batch_size)` with the reference's signatures;
  * batched: `train(n_envs=..., width=...)` — N self-play envs stepped by one kernel launch,
    2N transitions pushed per step, learn steps sampled from the device ring, gradients
    all-reduced over RCCL when launched with torch.distributed.run (one rank per GPU).
"""
import os
import random

import numpy as np
import torch
import torch.optim as optim

from config import *            # noqa: F401,F403
from config import BATCH_SIZE, GAMMA, MAP_WIDTH
from Net.DQNNet import Net

# DDQN.py:18-31
EPSILON_START = 1
ESPILON_END = 0.003
DECAY_RATE = 0.999
TAU = 0.001
MEM_CAPACITY = int(1e5)
UPDATE_EVERY = 4
GAME_CYCLE = 20
DISPLAY_CYCLE = GAME_CYCLE
folderName = 'survivor'


def planes_to_codes(planes):
    """pop_up planes [n, >=3, S, S] (wall, my, enemy) -> int8 observation codes [n, S, S]
    (inverse of util.pop_up, util.py:18-27): how float states handed to the drop-in
    ReplayBuffer.add are packed for the int8 ring."""
    wall, my, en = planes[:, 0], planes[:, 1], planes[:, 2]
    codes = torch.ones_like(wall, dtype=torch.int8)
    codes = torch.where(wall > 0.5, torch.full_like(codes, -1), codes)
    codes = torch.where(my > 5, torch.full_like(codes, 10), torch.where(my > 0.5, torch.full_like(codes, -2), codes))
    codes = torch.where(en > 5, torch.full_like(codes, -10), torch.where(en > 0.5, torch.full_like(codes, -3), codes))
    return codes


class ReplayBuffer:
    """DDQN.ReplayBuffer (DDQN.py:167-203) on a device ring (tron.vec.DeviceReplay)."""

    def __init__(self, action_size, buffer_size, batch_size, width=MAP_WIDTH, channels=4, plane4=0.0, seed=None,
                 rank=0):
        from tron.vec import DeviceReplay
        self.action_size = action_size
        self.batch_size = batch_size
        self.channels = channels
        self.plane4 = plane4
        self.side = width + 2
        self.memory = DeviceReplay(buffer_size, self.side * self.side,
                                   seed=random.getrandbits(32) if seed is None else seed, rank=rank)

    def add(self, state, action, reward, next_state, done):
        """One transition in the reference's types: float plane tensors (1,C,S,S), int, number, bool."""
        dev = self.memory.device
        s = planes_to_codes(torch.as_tensor(state, device=dev).reshape(1, -1, self.side, self.side))
        s2 = planes_to_codes(torch.as_tensor(next_state, device=dev).reshape(1, -1, self.side, self.side))
        self.memory.add(s, torch.tensor([int(action)], dtype=torch.int8, device=dev),
                        torch.tensor([float(reward)], dtype=torch.float32, device=dev), s2,
                        torch.tensor([int(bool(done))], dtype=torch.int8, device=dev))

    def add_batch(self, codes, actions, rewards, next_codes, dones):
        """codes None: the state rows were written ahead with self.memory.add_states()."""
        self.memory.add(codes, actions, rewards, next_codes, dones)

    def sample(self):                              # DDQN.py:191-200
        return self.memory.sample(self.batch_size, self.channels, self.plane4, side=self.side)

    def sample_codes(self):
        """sample() with the states left as the ring stores them: int8 observation codes [batch, S, S] (a twelfth of
        the bytes of the f32 planes).  Agent.learn takes either form."""
        return self.memory.sample_codes(self.batch_size, side=self.side)

    def __len__(self):
        return len(self.memory)


def _world(group=None):
    import torch.distributed as dist
    return dist.get_world_size(group) if (dist.is_available() and dist.is_initialized()) else 1


class AdamSoft(optim.Adam):
    """optim.Adam (DDQN.py:52: defaults, no weight decay / amsgrad) whose step() on device tensors is ONE launch that also applies
    Agent.soft_update to the target net's tensors (csrc/tron_dqn.hip::k_adam_soft; torch's fused form + the soft update: four
    launches, 75 us per learn step).  The state is torch's own (`step`, `exp_avg`, `exp_avg_sq` per parameter: state_dict() /
    load_state_dict() interchange with optim.Adam); `step` counters are host scalars.  Host tensors take optim.Adam's step."""

    def __init__(self, params, **kw):
        super().__init__(params, foreach=False, fused=False, capturable=False, **kw)

    @torch.no_grad()
    def step(self, closure=None, targets=None, tau=0.0):
        """targets: the target net's parameters, in the order of this optimizer's; theta_t <- tau theta + (1 - tau) theta_t after
        the step (DDQN.py:153-165).  Returns True when the soft update was applied here."""
        params = [p for g in self.param_groups for p in g["params"]]
        if not params or not params[0].is_cuda:
            super().step(closure)
            return False
        import ctypes as C
        from tron import _native as nat
        if len(self.param_groups) != 1:
            raise NotImplementedError("AdamSoft: one parameter group")
        grp = self.param_groups[0]
        if grp["weight_decay"] or grp["amsgrad"] or grp["maximize"]:
            raise NotImplementedError("AdamSoft: plain Adam only")
        targets = list(targets) if targets is not None else None
        if targets is not None and len(targets) != len(params):
            raise ValueError("AdamSoft.step: one target tensor per parameter")
        n = len(params)
        P, G, M, V, T = ((C.c_void_p * n)() for _ in range(5))
        numel, steps = (C.c_int64 * n)(), (C.c_double * n)()
        for k, p in enumerate(params):
            if p.dtype != torch.float32 or not p.is_contiguous():
                raise TypeError("AdamSoft: contiguous f32 parameters")
            P[k], numel[k] = p.data_ptr(), p.numel()
            if targets is not None:
                if targets[k].shape != p.shape or targets[k].dtype != torch.float32 or not targets[k].is_contiguous():
                    raise TypeError("AdamSoft: targets must match the parameters")
                T[k] = targets[k].data_ptr()
            if p.grad is None:
                continue
            g = p.grad
            if g.dtype != torch.float32 or not g.is_contiguous() or g.is_sparse:
                raise TypeError("AdamSoft: contiguous dense f32 gradients")
            st = self.state[p]
            if len(st) == 0:                                             # optim.Adam._init_group
                st["step"] = torch.tensor(0.0, dtype=torch.float32)
                st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            if st["step"].is_cuda:                                       # (a checkpoint of the fused form keeps them on the device)
                st["step"] = st["step"].detach().cpu()
            st["step"] += 1
            G[k], M[k], V[k], steps[k] = g.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(), float(st["step"])
        b1, b2 = grp["betas"]
        nat.check(nat.lib().tron_adam_soft_update(n, P, G, M, V, T if targets is not None else None, numel, steps, float(grp["lr"]),
                                                  float(b1), float(b2), float(grp["eps"]), float(tau), nat.stream_ptr()), "tron_adam_soft_update")
        return targets is not None


def all_reduce_mean_(flat, group=None):
    """flat <- mean over the ranks of flat, in place: ONE collective for the whole gradient (2.0 MB at 12x12, 4.6 MB at
    26x26: latency-bound on xGMI, so a single bucket).  RCCL for device tensors; the gloo rehearsal backend (several ranks
    sharing one GPU) reduces through host memory.  No-op without torch.distributed."""
    import torch.distributed as dist
    world = _world(group)
    if world == 1:
        return
    if dist.get_backend(group) == "gloo" and flat.is_cuda:
        host = flat.cpu()
        dist.all_reduce(host, op=dist.ReduceOp.SUM, group=group)
        flat.copy_(host)
    else:
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    flat /= world


class _TDLoss(torch.autograd.Function):
    """The Double-DQN loss of DDQN.py:129-146 (gather, arg-max, gather, target, MSELoss) and its gradient at the local net's
    Q-values as one launch of csrc/tron_dqn.hip instead of fourteen small ones."""

    @staticmethod
    def forward(ctx, q, actions, rewards, dones, ql_next, qt_next, gamma):
        from tron import _native as nat
        q = q.contiguous()
        loss = torch.empty((), dtype=torch.float32, device=q.device)
        grad_q = torch.empty_like(q)
        with torch.cuda.device(q.device):
            nat.check(nat.lib().tron_ddqn_td_loss(nat.ptr(q), nat.ptr(actions), nat.ptr(rewards), nat.ptr(dones), nat.ptr(ql_next),
                                                  nat.ptr(qt_next), float(gamma), q.shape[0], nat.ptr(loss), nat.ptr(grad_q),
                                                  nat.stream_ptr()), "tron_ddqn_td_loss")
        ctx.save_for_backward(grad_q)
        return loss

    @staticmethod
    def backward(ctx, grad_loss):
        (grad_q,) = ctx.saved_tensors
        return grad_q * grad_loss, None, None, None, None, None, None


def _td_fusable(q, actions, rewards, dones, ql_next, qt_next):
    return (q.is_cuda and q.dim() == 2 and q.shape[1] == 4 and q.dtype == torch.float32 and actions.dtype == torch.int64
            and rewards.dtype == torch.float32 and dones.dtype == torch.float32 and ql_next.shape == q.shape == qt_next.shape
            and all(t.is_contiguous() for t in (actions, rewards, dones, ql_next, qt_next))
            and actions.numel() == rewards.numel() == dones.numel() == q.shape[0] and q.shape[0] > 0
            and os.environ.get("TRON_TD_FUSED", "1") != "0")


def average_gradients(model, group=None):
    """The model's gradients averaged over the ranks.  When every .grad is a view of one flat buffer (Agent.flat_grads:
    what the trainer sets up) that buffer is reduced as it is — no cat, no copy back; otherwise the gradients are
    gathered into a temporary and scattered back."""
    if _world(group) == 1:
        return
    grads = [p.grad for p in model.parameters() if p.grad is not None]
    flat = getattr(model, "_tron_flat_grads", None)
    if flat is not None and sum(g.numel() for g in grads) == flat.numel() and all(
            g.untyped_storage().data_ptr() == flat.untyped_storage().data_ptr() for g in grads):
        all_reduce_mean_(flat, group)
        return
    flat = torch.cat([g.reshape(-1) for g in grads])
    all_reduce_mean_(flat, group)
    off = 0
    for g in grads:
        n = g.numel()
        g.copy_(flat[off:off + n].view_as(g))
        off += n


class Agent():
    def __init__(self, width=MAP_WIDTH, in_channels=4, device=None, buffer_size=MEM_CAPACITY, batch_size=BATCH_SIZE,
                 seed=None, rank=0, make_memory=True):
        self.device = torch.device(device if device is not None else ('cuda' if torch.cuda.is_available() else 'cpu'))
        self.qnetwork_local = Net(in_channels, width).to(self.device)            # DDQN.py:45-46
        self.qnetwork_target = Net(in_channels, width).to(self.device)
        self.action_size = 4
        self.steps = 0
        # DDQN.py:52.  On the GPU the update of all 22 tensors AND the soft update of the target net are one launch (AdamSoft; the
        # per-tensor loop is 22 x 5 small kernels, torch's fused form + a multi-tensor soft update four); TRON_ADAM_FUSED=0 keeps
        # torch's multi-tensor form.
        fused = self.device.type == "cuda" and os.environ.get("TRON_ADAM_FUSED", "1") != "0"
        self.optimizer = AdamSoft(self.qnetwork_local.parameters()) if fused else optim.Adam(self.qnetwork_local.parameters())
        self.epsilon = 0
        # act_batch's exploration draws: Philox keyed (seed, rank) at counter (observation, call).  The reference's
        # random.random() (DDQN.py:105) is unseeded, so without a seed the key comes from the OS; the call counter is part of
        # the full checkpoint (a resumed run continues the draw sequence instead of replaying it).
        self._eps_seed = int.from_bytes(os.urandom(4), "little") if seed is None else int(seed)
        self._eps_rank, self._eps_calls = int(rank), 0
        self._pending = self._pending_on_side = False                            # learn(defer=True) ... finish_learn()
        self._side = None
        self.totalloss = 0
        self.batch_size = batch_size
        self.memory = (ReplayBuffer(4, buffer_size, batch_size, width, in_channels, seed=seed, rank=rank)
                       if make_memory else None)
        self.t_step = 0
        for name, net in (('local_ai.bak', self.qnetwork_local), ('target_ai.bak', self.qnetwork_target)):
            path = 'ais/' + folderName + '/' + name                              # DDQN.py:61-64
            if os.path.isfile(path):
                net.load_state_dict(torch.load(path, map_location=self.device, weights_only=True))

    def get_loss(self):                            # DDQN.py:66-71
        out_loss = self.totalloss / max(self.steps, 1)
        self.totalloss = 0
        self.steps = 0
        return out_loss

    def step(self, state, action, reward, next_step, done):                      # DDQN.py:73-88
        self.memory.add(state, action, reward, next_step, done)
        self.t_step = (self.t_step + 1) % UPDATE_EVERY
        if self.t_step == 0 and len(self.memory) > self.batch_size:
            self.steps += 1
            self.learn(self.memory.sample(), GAMMA)

    def action(self, game_map):                    # DDQN.py:90-110
        self.qnetwork_local.eval()
        with torch.no_grad():
            action_values = self.qnetwork_local(game_map.to(self.device))
        self.qnetwork_local.train()
        if random.random() > self.epsilon:
            return int(np.argmax(action_values.cpu().data.numpy()))
        return int(random.choice(np.arange(self.action_size)))

    def act_batch(self, obs, epsilon, codes=False):
        """Batched epsilon-greedy on the device: obs f32 planes [B,C,S,S] — or, codes=True, the env's int8
        observation codes [B,S,S] — -> int8 actions [B].  The greedy forward runs on Net.infer (HIP conv kernels)."""
        greedy = self.qnetwork_local.infer(obs, codes=codes, greedy=True)
        if (greedy.is_cuda and greedy.dtype == torch.int8 and greedy.is_contiguous() and torch.is_tensor(epsilon) and epsilon.is_cuda
                and epsilon.dtype == torch.float32 and epsilon.numel() == 1):
            # one launch (csrc/tron_dqn.hip): Philox draws keyed by (seed, rank) at counter (observation, call); epsilon stays on the device
            from tron import _native as nat
            out = torch.empty_like(greedy)
            self._eps_calls += 1
            with torch.cuda.device(greedy.device):
                nat.check(nat.lib().tron_eps_greedy(nat.ptr(greedy), greedy.numel(), nat.ptr(epsilon.reshape(1)), self._eps_seed & 0xFFFFFFFF,
                                                    self._eps_rank, self._eps_calls, nat.ptr(out), nat.stream_ptr()), "tron_eps_greedy")
            return out
        rnd = torch.randint(0, self.action_size, greedy.shape, device=greedy.device, dtype=torch.int8)
        explore = torch.rand(greedy.shape, device=greedy.device) <= epsilon
        return torch.where(explore, rnd, greedy)

    def targets(self, rewards, next_state, dones, gamma, plane4=0.0):
        """Double-DQN labels (DDQN.py:129-142): a* = argmax Q_local(s'), y = r + g Q_target(s', a*)(1-done).
        Both forwards are gradient-free and in eval mode: Net.infer.  next_state: f32 planes, or int8 observation codes
        [B, S, S] (then both forwards run the weight-stationary chain of csrc/tron_conv_ws.hip)."""
        codes = next_state.dtype == torch.int8
        actions_q_local = self.qnetwork_local.infer(next_state, codes=codes, plane4=plane4).max(1)[1].unsqueeze(1).long()
        labels_next = self.qnetwork_target.infer(next_state, codes=codes, plane4=plane4).gather(1, actions_q_local)
        return rewards + (gamma * labels_next * (1 - dones))

    def flat_grads(self):
        """One flat f32 buffer with every parameter's .grad a view into it (allocated once): the gradient all-reduce
        is then a single collective on memory that is already contiguous, and zeroing the gradients is one fill."""
        net = self.qnetwork_local
        flat = getattr(net, "_tron_flat_grads", None)
        params = list(net.parameters())
        if flat is None or flat.numel() != sum(p.numel() for p in params) or flat.device != params[0].device:
            flat = torch.zeros(sum(p.numel() for p in params), dtype=torch.float32, device=params[0].device)
            net._tron_flat_grads = flat
        off = 0
        for p in params:
            n = p.numel()
            if p.grad is None or p.grad.untyped_storage().data_ptr() != flat.untyped_storage().data_ptr():
                p.grad = flat[off:off + n].view_as(p)
            off += n
        return flat

    def finish_learn(self):
        """The second half of a learn(..., defer=True): wait for the gradient all-reduce (on the side stream, when it went
        out there), then optimizer step and soft update.  No-op when nothing is pending."""
        if not getattr(self, "_pending", False):
            return
        if getattr(self, "_pending_on_side", False):
            torch.cuda.current_stream(self.device).wait_stream(self._side)
        if not (isinstance(self.optimizer, AdamSoft) and self.optimizer.step(targets=self.qnetwork_target.parameters(), tau=TAU)):
            if not isinstance(self.optimizer, AdamSoft):
                self.optimizer.step()
            self.soft_update(self.qnetwork_local, self.qnetwork_target, TAU)
        self._pending = self._pending_on_side = False

    def learn(self, experiences, gamma, defer=False):           # DDQN.py:115-151
        """experiences: ReplayBuffer.sample() (f32 planes, as the reference hands them over) or .sample_codes() (int8
        codes: conv1 reads them directly, forward and target forwards alike).

        One rank per GPU (torch.distributed initialised): the flat gradient is all-reduced on a side stream.  With
        defer=True the call returns once that collective is queued and `finish_learn()` applies the update — the
        trainer calls it after it has queued the next policy forward, so the collective's latency (xGMI ring: tens of
        microseconds for 2-5 MB) hides under that forward; the policy then acts on weights one learn step old, on every
        rank alike.  defer=False (what a single process runs): the reference's order, update applied before returning.
        (defer=True is honoured in a single process and on host tensors too — the gradients are then simply kept until
        finish_learn() — which is how the CPU tests pin the deferred order against a plain run.)"""
        self.finish_learn()
        states, actions, rewards, next_state, dones = experiences
        criterion = torch.nn.MSELoss()
        self.qnetwork_local.train()
        self.qnetwork_target.eval()
        plane4 = self.memory.plane4 if self.memory is not None else 0.0
        q_all = self.qnetwork_local.forward_codes(states, plane4) if states.dtype == torch.int8 else self.qnetwork_local(states)
        loss = None
        if q_all.is_cuda:
            codes = next_state.dtype == torch.int8
            ql_next = self.qnetwork_local.infer(next_state, codes=codes, plane4=plane4)
            qt_next = self.qnetwork_target.infer(next_state, codes=codes, plane4=plane4)
            if _td_fusable(q_all, actions, rewards, dones, ql_next, qt_next):
                loss = _TDLoss.apply(q_all, actions, rewards, dones, ql_next, qt_next, gamma)
            else:
                labels = rewards + (gamma * qt_next.gather(1, ql_next.max(1)[1].unsqueeze(1).long()) * (1 - dones))
                loss = criterion(q_all.gather(1, actions), labels)
        if loss is None:
            loss = criterion(q_all.gather(1, actions), self.targets(rewards, next_state, dones, gamma, plane4))
        self.totalloss += loss.detach()
        # One rank per GPU: every .grad is a view of one flat buffer (one fill, one collective).  A single process has
        # nothing to reduce, and a .grad that exists makes autograd ACCUMULATE into it — one small add kernel per
        # parameter, 24 launches of ~5 us per learn step — so there the gradients are dropped and autograd assigns them.
        if self.device.type == "cuda" and (_world() > 1 or getattr(self, "force_flat_grads", False)):
            flat = self.flat_grads()
            flat.zero_()
        else:
            flat = None
            self.optimizer.zero_grad(set_to_none=True)
        loss.backward()
        if flat is not None and _world() > 1:      # RCCL over xGMI when run one-rank-per-GPU
            if getattr(self, "_side", None) is None:
                self._side = torch.cuda.Stream(self.device)
            cur = torch.cuda.current_stream(self.device)
            self._side.wait_stream(cur)
            with torch.cuda.stream(self._side):
                all_reduce_mean_(flat)
            self._pending_on_side = True
        else:
            average_gradients(self.qnetwork_local)     # host tensors / gathered gradients: synchronous; no-op in a single process
        self._pending = True
        if not defer:
            self.finish_learn()
        return loss.detach()

    def soft_update(self, local_model, target_model, tau):                       # DDQN.py:153-165
        # theta_t <- tau theta + (1 - tau) theta_t over all 22 tensors in two multi-tensor launches (the per-tensor loop
        # of the reference is 88 small launches per learn step)
        with torch.no_grad():
            tp, lp = list(target_model.parameters()), list(local_model.parameters())
            if tp and tp[0].is_cuda:
                torch._foreach_mul_(tp, 1 - tau)
                torch._foreach_add_(tp, lp, alpha=tau)
            else:
                for target_param, local_param in zip(tp, lp):
                    target_param.data.copy_(tau * local_param.data + (1 - tau) * target_param.data)


def save_checkpoint(path, brain, epsilon=0.0, counters=None, replay="cursor"):
    """Everything needed to resume: both nets, Adam state, epsilon, counters, the exploration draw counter and the replay
    ring — replay="cursor": write head, fill level and sampler call counter (SURVEY 8(f)4's "replay head"); "contents":
    the filled slots too (2 * cells + 6 bytes each: 1.35 GB for 1 M slots at 24x24 boards); None: nothing.  Tensors and
    plain numbers only: the file loads with weights_only=True.  (The reference saves only the target net's state_dict,
    DDQN.py:326, so a resumed run restarts epsilon at 1 with an empty deque; `torch.save(brain.qnetwork_target.state_dict(), ...)`
    still gives that file.)"""
    ck = {"local": brain.qnetwork_local.state_dict(), "target": brain.qnetwork_target.state_dict(),
          "optimizer": brain.optimizer.state_dict(), "epsilon": float(epsilon), "t_step": brain.t_step,
          "eps_calls": int(brain._eps_calls), "eps_seed": int(brain._eps_seed), "counters": dict(counters or {})}
    ring = getattr(getattr(brain, "memory", None), "memory", None)
    if replay is not None and ring is not None and hasattr(ring, "state_dict"):
        ck["replay"] = ring.state_dict(contents=(replay == "contents"))
    torch.save(ck, path)


def load_checkpoint(path, brain):
    """Inverse of save_checkpoint.  A replay section with contents refills the ring; a cursor-only one is applied when the
    ring already holds that many transitions and skipped otherwise (a fresh ring restarts empty, as the reference's does)."""
    ck = torch.load(path, map_location="cpu", weights_only=True)
    brain.qnetwork_local.load_state_dict(ck["local"])
    brain.qnetwork_target.load_state_dict(ck["target"])
    brain.optimizer.load_state_dict(ck["optimizer"])
    brain.t_step = ck["t_step"]
    brain._eps_calls = int(ck.get("eps_calls", 0))
    if "eps_seed" in ck:
        brain._eps_seed = int(ck["eps_seed"])
    ring = getattr(getattr(brain, "memory", None), "memory", None)
    if "replay" in ck and ring is not None and hasattr(ring, "load_state_dict"):
        sd = ck["replay"]
        if "states" in sd or int(sd["size"]) <= len(ring):
            ring.load_state_dict(sd)
    return ck["epsilon"], ck["counters"]


def abort_job(exc=None, code=1):
    """One rank per GPU: end THIS process non-zero, now.  A rank that raised has left the per-learn-step all-reduce; its
    peers would block in RCCL until a watchdog fires (and a caller that swallowed the exception would report success).
    The traceback is printed, the process group is aborted where the backend can (NCCL / RCCL communicators: a blocked peer
    then fails instead of waiting) and the process exits with `code` without running atexit handlers that could block in
    the same collective — never a re-exec.  The launcher (`torch.distributed.run`) sees the failed worker and stops the
    others; peers on gloo see the closed connection and raise.  Reference shape: independent workers, ACKTR.py:183,285-289."""
    import sys
    import threading
    import traceback
    if exc is not None:
        traceback.print_exception(type(exc), exc, exc.__traceback__, file=sys.stderr)
    print(f"[tron] rank {os.environ.get('RANK', '0')}: aborting the job (exit code {code})", file=sys.stderr, flush=True)
    sys.stdout.flush()
    threading.Timer(5.0, lambda: os._exit(code)).start()       # if the abort below blocks, leave anyway
    try:
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            abort = getattr(dist.distributed_c10d, "_abort_process_group", None)
            if abort is not None and dist.get_backend() == "nccl":
                abort()
    except Exception:
        pass
    os._exit(code)


def fails_the_job(fn):
    """Decorator for a rank's main loop: with world > 1 any exception (or KeyboardInterrupt) becomes abort_job(); a single
    process re-raises as usual."""
    import functools

    @functools.wraps(fn)
    def guarded(*args, **kwargs):
        try:
            return fn(*args, **kwargs)
        except BaseException as e:        # noqa: BLE001 - the point is that nothing survives silently
            if _world() > 1 and not isinstance(e, SystemExit):
                abort_job(e)
            raise
    return guarded


def _decays_left(epsilon):
    """How many more times the reference's rule `if eps * DECAY_RATE > ESPILON_END: eps *= DECAY_RATE`
    (DDQN.py:313-315) fires from `epsilon` on: the rule is monotone, so it is a count."""
    k = 0
    while epsilon * DECAY_RATE > ESPILON_END:
        epsilon *= DECAY_RATE
        k += 1
    return k


@fails_the_job
def train(n_envs=4096, width=MAP_WIDTH, steps=200, learn_every=2, batch_size=BATCH_SIZE, capacity=1 << 20,
          in_channels=3, seed=0x5EED, log_every=50, save_path=None, log_dir=None, resume=None,
          terminal_next_state=True, brain=None):
    """Batched self-play DDQN: the loop of DDQN.py:225-346 with N envs per launch.
    Honours the reference's cadence as defaults (App. A #12): one learn step per 2 env-steps
    (UPDATE_EVERY=4 counted in per-player `brain.step` calls), epsilon x0.999 per 20 finished
    games, target net saved.  Returns a dict of counters.

    Nothing in the loop reads the device back: the finished-game counter, the 20-game cycle count and
    epsilon live in device tensors (epsilon = eps0 * DECAY_RATE ** decays, the closed form of the
    reference's repeated multiply); the host reads them at log time and at the end only.

    terminal_next_state=True stores what the reference stores for a finished game (DDQN.py:265-308: the
    terminal board as next_state) by stepping without autoreset and restarting the finished envs with a
    masked reset; False uses the env's autoreset (ACKTR.py:307-310 convention: next_state of a terminal
    transition is the new game's first observation — irrelevant to the (1 - done) target, one launch fewer).

    Update order.  A single process applies every learn step before the next policy forward (the reference's order,
    DDQN.py:73-88).  With one rank per GPU (world > 1) the gradient all-reduce goes out on a side stream and the update is
    applied AFTER the next policy forward has been queued (learn(defer=True) ... act_batch ... finish_learn): the policy of
    env step t + 1 acts on the weights of learn step k - 1, on every rank alike — a deterministic one-step staleness that
    hides the collective's latency; tests/test_dist_cpu.py holds a two-rank run to a single process applying that order.

    Failure.  With world > 1 an exception on any rank ends THIS process non-zero at once (abort_job): the launcher then
    takes the job down instead of the peers blocking in the next all-reduce."""
    import time
    import torch.distributed as dist
    from tron.vec import VecTron, pop_up_planes
    rank = dist.get_rank() if dist.is_initialized() else 0
    world = dist.get_world_size() if dist.is_initialized() else 1
    defer = world > 1
    if brain is None:
        torch.manual_seed(seed)                    # same initial weights on every rank ...
        brain = Agent(width, in_channels, buffer_size=capacity, batch_size=batch_size, seed=seed, rank=rank,
                      make_memory=True)
    torch.manual_seed(seed + 0x9E3779B1 * (rank + 1))   # ... but rank-own exploration and dropout draws
    dev = brain.device
    env = VecTron(n_envs, width, mode=None, seed=seed, rank=rank, obs_format="codes", reward="ddqn")
    S = width + 2
    # mode None + int8 codes: the observation buffer IS the env state and every step overwrites it in place.  The state
    # part of a transition is therefore written into the ring BEFORE the step (add_states) and the rest after it —
    # no per-step clone of the 2N observations.
    codes = env.reset().reshape(2 * n_envs, S, S)
    in_place = env.obs_is_state
    if not in_place:
        codes = codes.clone()
    eps0 = float(EPSILON_START)
    if resume:
        eps0, _ = load_checkpoint(resume, brain)
    # device-side bookkeeping (no per-step host sync)
    games_d = torch.zeros((), dtype=torch.int64, device=dev)
    cycles_d = torch.zeros((), dtype=torch.int64, device=dev)
    decays_d = torch.zeros((), dtype=torch.int64, device=dev)
    decays_max = torch.tensor(_decays_left(eps0), dtype=torch.int64, device=dev)
    eps_d = torch.tensor(eps0, dtype=torch.float64, device=dev)
    decay_d = torch.tensor(DECAY_RATE, dtype=torch.float64, device=dev)
    # the same bookkeeping as ONE launch per env step (tron_eps_schedule) when the env is on the GPU: {games, cycles, decays, decays_max}
    sched = torch.tensor([0, 0, 0, _decays_left(eps0)], dtype=torch.int64, device=dev) if dev.type == "cuda" else None
    eps_f32 = torch.tensor([eps0], dtype=torch.float32, device=dev)
    learn_steps, transitions, games_seen = 0, 0, 0
    writer = None
    if log_dir and rank == 0:
        from tron.scalars import ScalarWriter
        writer = ScalarWriter(log_dir)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for it in range(steps):
        actions = brain.act_batch(codes, eps_f32 if sched is not None else eps_d.to(torch.float32), codes=True).reshape(n_envs, 2)   # conv1 reads the codes
        brain.finish_learn()                       # (one rank per GPU: the previous learn step's all-reduce ran under this forward)
        if in_place:
            brain.memory.memory.add_states(codes)
        obs, reward, done, _ = env.step(actions, autoreset=not terminal_next_state)
        next_codes = obs.reshape(2 * n_envs, S, S)
        brain.memory.add_batch(None if in_place else codes, actions.reshape(-1), reward.reshape(-1), next_codes,
                               done.repeat_interleave(2))
        transitions += 2 * n_envs
        if terminal_next_state:
            env.reset(mask=done)                   # finished games restart; obs now holds the new first states
        codes = env.obs.reshape(2 * n_envs, S, S)
        if not in_place:
            codes = codes.clone()
        if sched is None:
            games_d += done.sum()
        if it % learn_every == learn_every - 1 and len(brain.memory) > batch_size:    # len(): a host counter
            brain.steps += 1
            brain.learn(brain.memory.sample_codes(), GAMMA, defer=defer)  # the batch as int8 codes: conv1 and the target chain read them
            learn_steps += 1
        # DDQN.py:313-315, once per finished 20-game cycle
        if sched is not None:
            from tron import _native as nat
            d8 = done if done.dtype == torch.int8 else done.to(torch.int8)
            with torch.cuda.device(dev):
                nat.check(nat.lib().tron_eps_schedule(nat.ptr(d8.contiguous()), d8.numel(), nat.ptr(sched), GAME_CYCLE, eps0, DECAY_RATE,
                                                      nat.ptr(eps_d.reshape(1)), nat.ptr(eps_f32), nat.stream_ptr()), "tron_eps_schedule")
            games_d = sched[0]
        else:
            new_cycles = torch.div(games_d, GAME_CYCLE, rounding_mode="floor")
            decays_d = torch.minimum(decays_d + (new_cycles - cycles_d), decays_max)
            cycles_d = new_cycles
            eps_d = eps0 * torch.pow(decay_d, decays_d)
        if log_every and rank == 0 and it % log_every == log_every - 1:
            loss, games_seen, epsilon = float(brain.get_loss()), int(games_d), float(eps_d)
            print(f"step {it + 1}: games {games_seen} eps {epsilon:.4f} loss {loss:.4f}", flush=True)
            if writer:                                                            # DDQN.py:342-344
                writer.add_scalar('Training loss', loss, games_seen)
                writer.add_scalar('Duration', n_envs * (it + 1) / max(games_seen, 1), games_seen)
                writer.add_scalar('Epsilon', epsilon, games_seen)
    brain.finish_learn()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    games, epsilon = int(games_d), float(eps_d)
    if writer:
        writer.close()
    if save_path and rank == 0:
        torch.save(brain.qnetwork_target.state_dict(), save_path)                 # DDQN.py:326 saves the TARGET net
        save_checkpoint(save_path + ".full", brain, epsilon, dict(games=games, learn_steps=learn_steps))
    return dict(env_steps=n_envs * steps * world, transitions_pushed=transitions * world,
                transitions_learned=learn_steps * batch_size * world, learn_steps=learn_steps, games=games * world,
                seconds=dt, env_steps_per_s=n_envs * steps * world / dt,
                learned_transitions_per_s=learn_steps * batch_size * world / dt, epsilon=epsilon, brain=brain)


def main():
    """`python DDQN.py` trained one hard-wired configuration in the reference (DDQN.py:206-349); the same
    loop here takes its sizes from the command line.  Under torch.distributed.run every rank trains its
    own env shard and replay shard and the gradients are averaged over RCCL."""
    import argparse
    import os
    import torch.distributed as dist
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", type=int, default=4096, help="parallel self-play games per GPU")
    ap.add_argument("--width", type=int, default=MAP_WIDTH)
    ap.add_argument("--steps", type=int, default=2000, help="env steps (each steps every game once)")
    ap.add_argument("--batch", type=int, default=BATCH_SIZE, help="learn batch (config.py:7)")
    ap.add_argument("--capacity", type=int, default=1 << 20, help="replay slots in HBM")
    ap.add_argument("--log-every", type=int, default=50)
    ap.add_argument("--log-dir", default=None, help="TensorBoard event files + scalars.jsonl")
    ap.add_argument("--save", default="save/DDQN.bak", help="target-net state_dict, like DDQN.py:326")
    ap.add_argument("--resume", default=None, help="full checkpoint written by save_checkpoint")
    a = ap.parse_args()
    if int(os.environ.get("WORLD_SIZE", "1")) > 1:
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")) % max(torch.cuda.device_count(), 1))
        dist.init_process_group(os.environ.get("TRON_DIST_BACKEND", "nccl"))
    if a.save:
        os.makedirs(os.path.dirname(a.save) or ".", exist_ok=True)
    out = train(n_envs=a.envs, width=a.width, steps=a.steps, batch_size=a.batch, capacity=a.capacity,
                log_every=a.log_every, save_path=a.save, log_dir=a.log_dir, resume=a.resume)
    print({k: v for k, v in out.items() if k != "brain"})
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
