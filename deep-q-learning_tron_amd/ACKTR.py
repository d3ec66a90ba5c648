"""A2C / ACKTR trainer — the reference's ACKTR.py (RolloutStorage :24-69, Brain :72-159,
train :162-437) with the env loop replaced by one VecTron launch per step.

* `RolloutStorage`, `Brain(actor_critic, args, acktr)` keep the reference's methods and maths;
  tensors live on the model's device instead of being shuffled through the host per step.
* `train(...)` is the loop of ACKTR.py:263-375 for N parallel self-play envs (both players act
  with the same network, two rollout storages, two updates per iteration), with the reference's
  conventions: reward -1 per step and get_reward(constants) at the end of a game; on done the
  env is replaced and the observation stored is the NEW game's (ACKTR.py:307-310); masks = 1 - done.
"""
import argparse
import os
import time

import torch
from torch import optim

from config import *            # noqa: F401,F403
import config
from config import GAMMA, MAP_WIDTH, NUM_ADVANCED_STEP, NUM_PROCESSES
from Net.ACNet import MapNet, Mulnet, TestNet  # noqa: F401
from Net.kfac import KFACOptimizer

folderName = 'save'


class RolloutStorage(object):
    """n-step rollout memory (ACKTR.py:24-69).  `env_dim` > 0 adds the per-step env vector
    (`probs`) that the non-MapNet nets take."""

    def __init__(self, num_steps, num_processes, channels=3, width=MAP_WIDTH, env_dim=2, device="cpu"):
        S = width + 2
        self.num_steps = num_steps
        self.observations = torch.zeros(num_steps + 1, num_processes, channels, S, S, device=device)
        self.masks = torch.ones(num_steps + 1, num_processes, 1, device=device)
        self.rewards = torch.zeros(num_steps, num_processes, 1, device=device)
        self.actions = torch.zeros(num_steps, num_processes, 1, device=device).long()
        self.probs = torch.zeros(num_steps, num_processes, env_dim, device=device).float() if env_dim else None
        self.returns = torch.zeros(num_steps + 1, num_processes, 1, device=device)
        self.index = 0

    def insert(self, current_obs, action, reward, mask, probs=None):
        self.observations[self.index + 1].copy_(current_obs)
        self.masks[self.index + 1].copy_(mask)
        self.rewards[self.index].copy_(reward)
        self.actions[self.index].copy_(action)
        if probs is not None:
            self.probs[self.index].copy_(probs)
        self.index = (self.index + 1) % self.num_steps

    def after_update(self):
        self.observations[0].copy_(self.observations[-1])
        self.masks[0].copy_(self.masks[-1])

    def compute_returns(self, next_value):
        """Discounted n-step returns, newest first (ACKTR.py:60-69)."""
        self.returns[-1] = next_value
        for ad_step in reversed(range(self.rewards.size(0))):
            self.returns[ad_step] = self.returns[ad_step + 1] * GAMMA * self.masks[ad_step + 1] + self.rewards[ad_step]


class Brain(object):
    def __init__(self, actor_critic, args=None, acktr=False, device=None):
        self.device = torch.device(device if device is not None else config.device)
        self.actor_critic = actor_critic.to(self.device)
        self.acktr = acktr
        p = getattr(args, "p", None)
        v = getattr(args, "v", None)
        self.policy_loss_coef = config.policy_loss_coef if p is None else float(p)
        self.value_loss_coef = config.value_loss_coef if v is None else float(v)
        # the sampled-Fisher backward pass computes statistics only (see update()); TRON_ACKTR_FISHER_FULL=1: as two full passes
        self.fisher_stats_only = os.environ.get("TRON_ACKTR_FISHER_FULL", "0") == "0"
        from Net import activations
        self._grad_scope = activations.GradScope()      # this Brain's graphs: the statistics-only switch reaches no other backward
        if acktr:
            self.optimizer = KFACOptimizer(self.actor_critic)
        else:
            self.optimizer = optim.RMSprop(self.actor_critic.parameters(), config.lr, eps=config.eps,
                                           alpha=config.alpha)

    def _cheap_parameters(self):
        """Every parameter except the weights whose gradient is a large product (convolutions over >= 16 input channels,
        Linear layers with >= 1024 inputs): biases, the first convolution, the small Linear layers.  Asking autograd for
        these reaches every module of the net (each module's output gradient is needed on the way down), which is all
        the statistics pass needs."""
        import torch.nn as nn
        heavy = set()                                   # (rebuilt per call — a few dozen modules: split_bias or a model swap may have replaced parameters)
        for m in self.actor_critic.modules():
            if isinstance(m, nn.Conv2d) and m.in_channels >= 16:
                heavy.add(id(m.weight))
            elif isinstance(m, nn.Linear) and m.in_features >= 1024:
                heavy.add(id(m.weight))
        return [p for p in self.actor_critic.parameters() if id(p) not in heavy and p.requires_grad]

    def update(self, rollouts, micro_batch=None):
        """One update from a full rollout (ACKTR.py:88-159).  With `micro_batch`, the T*N samples are
        pushed through the network in slices of that size: losses, gradients and K-FAC factor sums
        are accumulated with the weights that make the result equal to the single big batch."""
        num_steps, num_processes = rollouts.rewards.size(0), rollouts.rewards.size(1)
        B = num_steps * num_processes
        obs_shape = rollouts.observations.shape[2:]
        self.optimizer.zero_grad()
        obs = rollouts.observations[:-1].reshape(-1, *obs_shape)
        acts = rollouts.actions.view(-1, 1)
        probs = None if rollouts.probs is None else rollouts.probs.view(-1, rollouts.probs.size(-1))
        rets = rollouts.returns[:-1].reshape(-1, 1)
        fisher = self.acktr and self.optimizer.steps % self.optimizer.Ts == 0
        mb = B if not micro_batch else min(int(micro_batch), B)
        split = mb < B
        if split and self.acktr:
            self.optimizer.begin_accumulate(B)
        # value noise of the sampled Fisher: drawn from the HOST generator and moved, as the reference does (ACKTR.py:133-135) —
        # the same torch.manual_seed then gives the same update on either device (tests/test_gpu_benchbatch.py)
        noise = torch.randn(B, 1).to(self.device, non_blocking=True) if fisher else None
        sums = torch.zeros(5, device=self.device)     # value_loss, action_gain, entropy, logp, advantage
        grads = None
        for lo in range(0, B, mb):
            hi = min(lo + mb, B)
            w = (hi - lo) / B
            o = obs[lo:hi].to(self.device).detach()
            a = acts[lo:hi].to(self.device).detach()
            from Net import activations
            with activations.grad_scope(self._grad_scope):       # the nodes of this forward belong to this Brain's scope
                if probs is None:
                    values, action_log_probs, entropy = self.actor_critic.evaluate_actions(o, a)
                else:
                    values, action_log_probs, entropy = self.actor_critic.evaluate_actions(
                        o, a, probs[lo:hi].to(self.device).detach())
            advantages = rets[lo:hi].to(self.device).detach() - values
            value_loss = advantages.pow(2).mean()
            action_gain = (action_log_probs * advantages.detach()).mean()
            if fisher:
                # sampled Fisher (Martens 2014): statistics of the gradients of these two losses
                self.actor_critic.zero_grad()
                pg_fisher_loss = -action_log_probs.mean()
                sample_values = values + noise[lo:hi]
                vf_fisher_loss = -(values - sample_values.detach()).pow(2).mean()
                fisher_loss = pg_fisher_loss + vf_fisher_loss
                self.optimizer.acc_stats = True
                if self.fisher_stats_only:
                    # This pass exists for K-FAC's gradient statistics (the hooks on every module's output gradient);
                    # weight gradients are a by-product — and the expensive half of a backward pass.  Ask autograd only
                    # for the cheap parameters (every module's hook still fires: its input gradient is on the way to
                    # them), drop what that leaves in .grad, and let the loss pass below carry the Fisher loss as well.
                    self._grad_scope.skip_weight_gradients = True     # (the hand-written convolution nodes of THIS graph: Net/activations.py)
                    try:
                        (fisher_loss * w).backward(retain_graph=True, inputs=self._cheap_parameters())
                    finally:
                        self._grad_scope.skip_weight_gradients = False
                    self.actor_critic.zero_grad()
                else:
                    (fisher_loss * w).backward(retain_graph=True)
                self.optimizer.acc_stats = False
                # NB the reference does not clear .grad here (ACKTR.py:131-150): the Fisher loss's
                # gradient stays in and is added to the update's gradient.  Honoured (either way: d(fisher) + d(total)).
            # the reference's loss reads the coefficients from config, not from the Brain (ACKTR.py:147-148)
            total = (value_loss * config.value_loss_coef - action_gain * config.policy_loss_coef
                     - entropy * config.entropy_coef)
            if fisher and self.fisher_stats_only:
                ((total + fisher_loss) * w).backward()
            else:
                (total * w).backward()
            if split and fisher:                        # keep this slice's gradient: the next Fisher pass zeroes .grad
                g = [p.grad.detach().clone() for p in self.actor_critic.parameters()]
                grads = g if grads is None else [x.add_(y) for x, y in zip(grads, g)]
            sums += w * torch.stack([value_loss.detach(), action_gain.detach(), entropy.detach(),
                                     action_log_probs.detach().mean(), advantages.detach().mean()])
        if split:
            if grads is not None:                       # otherwise .grad accumulated across the slices by itself
                for p, g in zip(self.actor_critic.parameters(), grads):
                    p.grad = g
            if self.acktr:
                self.optimizer.end_accumulate()
        from DDQN import average_gradients             # one rank per GPU: the update of the global batch (no-op in a single process)
        average_gradients(self.actor_critic)
        self.optimizer.step()
        value_loss, action_gain, entropy, logp, radv = sums.unbind(0)
        total_loss = value_loss * config.value_loss_coef - action_gain * config.policy_loss_coef - entropy * config.entropy_coef
        return total_loss, value_loss, action_gain, entropy, logp, radv


def _rank_world():
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def _fails_the_job(fn):
    import functools

    @functools.wraps(fn)
    def guarded(*a, **k):
        from DDQN import fails_the_job
        return fails_the_job(fn)(*a, **k)
    return guarded


@_fails_the_job
def train(n_envs=NUM_PROCESSES, width=MAP_WIDTH, model="mul", reward="3", iterations=100, acktr=True,
          num_steps=NUM_ADVANCED_STEP, gamemode=None, seed=0x5EED, log_every=0, save_path=None, args=None,
          micro_batch=16384, act_batch=16384, ai_p1=True, ai_p2=True, trace=None):
    """Batched self-play ACKTR/A2C on VecTron.  Returns counters and the Brain.
    trace (tests): called as trace("step", it, step, actions) after every env step and trace("collected", it, rollouts, returns
    bootstrap values) once an iteration's rollouts are complete, before the two updates consume them.
    One rank per GPU (torch.distributed initialised): the reference runs independent workers (ACKTR.py:183,285-289); here rank
    r owns its own n_envs envs (Philox stream (seed, r)) and the ranks train ONE net — same initial weights, every update that of
    the global batch: gradients and K-FAC factor samples are averaged over the ranks (DDQN.average_gradients,
    KFACOptimizer._mean_over_ranks), so the weights stay identical without ever being broadcast again.  A rank that fails ends
    the job (DDQN.fails_the_job).
    ai_p1 / ai_p2 False seat MinimaxPlayer(2, "voronoi") there (ACKTR.py:176-177,286-287): that
    player's executed move comes from the search; like the reference, the rollout still stores
    the move the net sampled."""
    from tron.vec import VecTron
    gamemode = config.GAME_MODE if gamemode is None else gamemode
    constants = {"1": config.reward_cons1, "2": config.reward_cons2, "3": config.reward_cons3}[str(reward)]
    torch.manual_seed(seed)
    is_map = (model == "map")
    net = MapNet(width) if is_map else Mulnet(width)
    brain = Brain(net, args, acktr=acktr, device="cuda")
    dev = brain.device
    rank, world = _rank_world()
    if world > 1:
        import torch.distributed as dist
        for t in list(net.parameters()) + list(net.buffers()):            # (the same seed already gave the same weights; this makes it a fact)
            if dist.get_backend() == "gloo" and t.is_cuda:                  # (rehearsals: gloo's collectives take host tensors)
                h = t.data.cpu()
                dist.broadcast(h, 0)
                t.data.copy_(h)
            else:
                dist.broadcast(t.data, 0)
        torch.manual_seed(seed + 1000003 * (rank + 1))                      # action sampling, dropout and Fisher noise: each rank its own draws
    env = VecTron(n_envs, width, mode=gamemode, seed=seed, rank=rank, obs_format="planes4" if is_map else "planes3",
                  reward=dict(step=-1.0, win=float(constants[0]), lose=float(constants[1]), draw=0.0, step_is_index=0))
    ch = 4 if is_map else 3
    rollouts = [RolloutStorage(num_steps, n_envs, ch, width, 0 if is_map else 2, dev) for _ in range(2)]

    def env_vectors():
        st = env.state()
        deg = st["degree"].to(torch.float32)
        return [torch.stack([deg, st["weight"][:, p].to(torch.float32)], 1) for p in range(2)]   # get_multy(p)

    obs = env.reset()
    for p in range(2):
        rollouts[p].observations[0].copy_(obs[:, p])
    probs = env_vectors()
    games, t0 = 0, time.perf_counter()
    stats = None
    update_events = []                             # (start, end) of the two Brain.update calls of every iteration
    for it in range(iterations):
        for step in range(num_steps):
            if not is_map:
                probs = env_vectors()
            with torch.no_grad():
                acts = []
                for p in range(2):
                    o = rollouts[p].observations[step]
                    acts.append(torch.cat([net.act(o[i:i + act_batch]) if is_map
                                           else net.act(o[i:i + act_batch], probs[p][i:i + act_batch])
                                           for i in range(0, n_envs, act_batch)]))
            actions = torch.cat(acts, 1).to(torch.int8)
            for p, is_ai in enumerate((ai_p1, ai_p2)):
                if not is_ai:
                    actions[:, p] = env.minimax_actions(p + 1)
            obs, reward, done, _ = env.step(actions, autoreset=True)
            masks = (1 - done.to(torch.float32)).unsqueeze(1)
            games += int(done.sum())
            for p in range(2):
                rollouts[p].insert(obs[:, p], acts[p], reward[:, p:p + 1], masks, None if is_map else probs[p])
            if trace is not None:
                trace("step", it, step, actions)
        with torch.no_grad():
            nxt = []
            for p in range(2):
                o = rollouts[p].observations[-1]
                nxt.append(torch.cat([net.get_value(o[i:i + act_batch]) if is_map
                                      else net.get_value(o[i:i + act_batch], probs[p][i:i + act_batch])
                                      for i in range(0, n_envs, act_batch)]))
        for p in range(2):
            rollouts[p].compute_returns(nxt[p])
        if trace is not None:
            trace("collected", it, rollouts, nxt)
        ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        ev[0].record()
        stats = brain.update(rollouts[0], micro_batch)
        brain.update(rollouts[1], micro_batch)
        ev[1].record()
        update_events.append(ev)
        for p in range(2):
            rollouts[p].after_update()
        if log_every and it % log_every == log_every - 1:
            print(f"iter {it + 1}: games {games} loss {float(stats[0]):.4f} value {float(stats[1]):.4f} "
                  f"entropy {float(stats[3]):.4f}", flush=True)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if save_path:
        os.makedirs(os.path.dirname(save_path) or ".", exist_ok=True)
        torch.save(brain.actor_critic.state_dict(), save_path)          # ACKTR.py:399
    upd = sum(a.elapsed_time(b) for a, b in update_events) / 1e3
    return dict(iterations=iterations, env_steps=iterations * num_steps * n_envs, games=games, seconds=dt,
                env_steps_per_s=iterations * num_steps * n_envs / dt, updates=2 * iterations, brain=brain,
                update_seconds=upd, rollout_seconds=dt - upd,
                last_stats=None if stats is None else [float(s) for s in stats])


def main():
    parser = argparse.ArgumentParser()
    parser.add_argument('-m', required=False, help='model structure: map | mul', default="mul")
    parser.add_argument('-r', required=False, help='reward condition number', default="3")
    parser.add_argument('-p', required=False, help='policy coefficient', default="0.7")
    parser.add_argument('-v', required=False, help='value coefficient', default="0.9")
    parser.add_argument('-u', required=False, help='unique string', default='multi_test')
    parser.add_argument('--envs', type=int, default=NUM_PROCESSES)
    parser.add_argument('--width', type=int, default=MAP_WIDTH)
    parser.add_argument('--iterations', type=int, default=1000)
    a = parser.parse_args()
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:                                   # one rank per GPU (python -m torch.distributed.run --nproc-per-node N ACKTR.py ...)
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")) % max(torch.cuda.device_count(), 1))
        dist.init_process_group(os.environ.get("TRON_DIST_BACKEND", "nccl"))
    rank = dist.get_rank() if world > 1 else 0
    out = train(a.envs, a.width, a.m, a.r, a.iterations, log_every=20 if rank == 0 else 0, args=a,
                save_path=(f"{folderName}/ACKTR_player{a.m}{a.u}.bak" if a.u and rank == 0 else None))   # (the ranks hold the same weights)
    if rank == 0:
        print({k: v for k, v in out.items() if k != "brain"})
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
