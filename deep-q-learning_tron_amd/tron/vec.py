"""VecTron — N independent TRON games stepped by one HIP kernel launch.

The batched counterpart of the reference's `envs = [make_game(...)] * N` list and
its `for i in range(N): envs[i].step(a1, a2)` loop (ACKTR.py:183,285-317).  All
state lives in HBM behind a `tron_handle` (include/tron_hip.h); this class only
owns the output tensors and launches kernels on torch's current stream.
"""
import ctypes as C

import torch

from . import _native as nat

# reward tables of the three reference trainers
REWARDS = {
    "ddqn": dict(step=-1.0, win=100.0, lose=-100.0, draw=0.0, step_is_index=0),   # DDQN.py:289-305
    "dqn": dict(step=0.0, win=100.0, lose=-25.0, draw=0.0, step_is_index=1),      # DQN.py:224-241
    "acktr": dict(step=-1.0, win=10.0, lose=-10.0, draw=0.0, step_is_index=0),    # ACKTR.py:294-317 + config.py:37
}


class VecTron:
    def __init__(self, n_envs, width=10, mode=None, fair=False, seed=0x5EED, rank=0, device=None,
                 obs_format="codes", reward="ddqn", slide=None, obs_is_state=True, incremental=False):
        if not torch.cuda.is_available():
            raise nat.TronNativeError("VecTron needs a HIP device (no CPU fallback)")
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self.N, self.W = int(n_envs), int(width)
        self.S = self.W + 2
        self.G = self.S * self.S
        self.mode = mode
        self.obs_format = obs_format
        self._fmt = nat.OBS[obs_format]
        self._lib = nat.lib()
        h = C.c_void_p()
        with torch.cuda.device(self.device):
            nat.check(self._lib.tron_create(self.N, self.W, nat.MODE[mode], int(bool(fair)), seed & 0xFFFFFFFF,
                                            rank & 0xFFFFFFFF, C.byref(h)), "tron_create")
        self._h = h
        self.set_reward(**(REWARDS[reward] if isinstance(reward, str) else reward))
        if slide is not None:
            self.set_slide(slide)
        dev = self.device
        self.obs = self._alloc_obs(self._fmt)
        self.done = torch.zeros(self.N, dtype=torch.int8, device=dev)
        self.winner = torch.zeros(self.N, dtype=torch.int8, device=dev)
        self.reward = torch.zeros(self.N, 2, dtype=torch.float32, device=dev)
        # int8 codes + even side: let self.obs BE the env state (tron_attach_obs_state); it is then read-only for the
        # caller — clone what must outlive the next step.  (The sliding modes too: their slide tiles, which the codes show
        # as bodies, are kept in a per-env log for grid().)
        self.obs_is_state = bool(obs_is_state and self._fmt == nat.OBS_CODES_I8 and self.W % 2 == 0)
        # incremental=True (needs obs_is_state, mode None): steps write only the cells a move touches and the boards
        # that restart, instead of rewriting both planes — same observations, far less traffic
        self.incremental = bool(incremental and self.obs_is_state and mode in (None, "none"))
        if self.obs_is_state:
            with torch.cuda.device(self.device):
                nat.check(self._lib.tron_attach_obs_state(self._h, nat.ptr(self.obs), nat.stream_ptr()),
                          "tron_attach_obs_state")

    # -- plumbing ---------------------------------------------------------
    def _alloc_obs(self, fmt):
        N, S = self.N, self.S
        if fmt == nat.OBS_NONE:
            return None
        if fmt == nat.OBS_CODES_I8:
            return torch.empty(N, 2, S, S, dtype=torch.int8, device=self.device)
        ch = 3 if fmt == nat.OBS_PLANES3_F32 else 4
        return torch.empty(N, 2, ch, S, S, dtype=torch.float32, device=self.device)

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self._lib.tron_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _dev_arg(self, t, dtype, shape):
        if t is None:
            return None
        t = torch.as_tensor(t, device=self.device).to(dtype).contiguous()
        if tuple(t.shape) != tuple(shape):
            raise ValueError(f"expected shape {tuple(shape)}, got {tuple(t.shape)}")
        return t

    # -- configuration ----------------------------------------------------
    def set_reward(self, step=-1.0, win=100.0, lose=-100.0, draw=0.0, step_is_index=0):
        nat.check(self._lib.tron_set_reward(self._h, step, win, lose, draw, int(step_is_index)), "tron_set_reward")

    def set_slide(self, slide):
        """Game(..., slide_pram=slide) (game.py:88); a float or a float64 [N] tensor."""
        with torch.cuda.device(self.device):
            if torch.is_tensor(slide):
                t = self._dev_arg(slide, torch.float64, (self.N,))
                nat.check(self._lib.tron_set_slide(self._h, 0.0, nat.ptr(t), nat.stream_ptr()), "tron_set_slide")
            else:
                nat.check(self._lib.tron_set_slide(self._h, float(slide), None, nat.stream_ptr()), "tron_set_slide")

    def set_weight_degree(self, weight=None, degree=None):
        """Assign Game.weight / Game.degree (game.py:83,87): int16 [N,2] / [N]."""
        w = self._dev_arg(weight, torch.int16, (self.N, 2))
        d = self._dev_arg(degree, torch.int16, (self.N,))
        with torch.cuda.device(self.device):
            nat.check(self._lib.tron_set_weight_degree(self._h, nat.ptr(w), nat.ptr(d), nat.stream_ptr()),
                      "tron_set_weight_degree")

    # -- reset / step / encode --------------------------------------------
    def reset(self, mask=None, start_pos=None, weight=None, degree=None):
        """make_game for the masked envs (all when mask is None); returns the observation."""
        m = self._dev_arg(mask, torch.int8, (self.N,))
        sp = self._dev_arg(start_pos, torch.int8, (self.N, 4))
        w = self._dev_arg(weight, torch.int16, (self.N, 2))
        d = self._dev_arg(degree, torch.int16, (self.N,))
        with torch.cuda.device(self.device):
            nat.check(self._lib.tron_reset(self._h, nat.ptr(m), nat.ptr(sp), nat.ptr(w), nat.ptr(d),
                                           nat.stream_ptr()), "tron_reset")
        return self.encode() if self._fmt != nat.OBS_NONE else None

    def step(self, actions=None, uniforms=None, autoreset=True, nonreversing=False):
        """One Game.step for every env.  actions int8 [N,2] in 0..3 (None: i.i.d. uniform
        from the env's Philox stream, or — nonreversing=True — uniform over the three headings
        that do not reverse the player's last move); uniforms f32 [N,2] for ice/temper (None: Philox).
        Returns (obs, reward, done, winner) — tensors owned by this object, overwritten by
        the next call."""
        a = self._dev_arg(actions, torch.int8, (self.N, 2))
        u = self._dev_arg(uniforms, torch.float32, (self.N, 2))
        flags = ((nat.STEP_AUTORESET if autoreset else 0) | (nat.STEP_INCREMENTAL if self.incremental else 0) |
                 (nat.STEP_NONREVERSING if nonreversing else 0))
        with torch.cuda.device(self.device):
            nat.check(self._lib.tron_step_encode(self._h, nat.ptr(a), nat.ptr(u), flags, self._fmt,
                                                 nat.ptr(self.obs), nat.ptr(self.done), nat.ptr(self.winner),
                                                 nat.ptr(self.reward), nat.stream_ptr()), "tron_step_encode")
        return self.obs, self.reward, self.done, self.winner

    def part_range(self, part, nparts):
        """(first_env, n_envs) of slice `part` of `nparts` (whole tiles of consecutive envs, split evenly)."""
        a, b = C.c_int32(), C.c_int32()
        nat.check(self._lib.tron_part_range(self._h, int(part), int(nparts), C.byref(a), C.byref(b)), "tron_part_range")
        return a.value, b.value

    def step_part(self, part, nparts, actions=None, uniforms=None, autoreset=True, nonreversing=False):
        """step() for one slice of the envs, on torch's current stream: run the slices as independent pipelines
        (one stream each) so that the policy forward of one slice overlaps the env kernel of another.  `actions`
        / `uniforms` and the returned tensors are the FULL [N, ...] buffers; only the slice's rows are read and
        written (part_range tells which)."""
        a = self._dev_arg(actions, torch.int8, (self.N, 2))
        u = self._dev_arg(uniforms, torch.float32, (self.N, 2))
        flags = (nat.STEP_AUTORESET if autoreset else 0) | (nat.STEP_NONREVERSING if nonreversing else 0)
        with torch.cuda.device(self.device):
            nat.check(self._lib.tron_step_encode_part(self._h, int(part), int(nparts), nat.ptr(a), nat.ptr(u), flags,
                                                      self._fmt, nat.ptr(self.obs), nat.ptr(self.done),
                                                      nat.ptr(self.winner), nat.ptr(self.reward), nat.stream_ptr()),
                      "tron_step_encode_part")
        return self.obs, self.reward, self.done, self.winner

    def step_fn(self, autoreset=True, nonreversing=False):
        """A zero-argument callable that launches one random-action step (Philox actions) with all
        ctypes arguments bound once — for launch loops where Python argument handling per call
        would otherwise dominate a ~25 us kernel.  Same outputs as step()."""
        fn = self._lib.tron_step_encode
        flags = ((nat.STEP_AUTORESET if autoreset else 0) | (nat.STEP_INCREMENTAL if self.incremental else 0) |
                 (nat.STEP_NONREVERSING if nonreversing else 0))
        args = (self._h, None, None, flags, self._fmt, nat.ptr(self.obs),
                nat.ptr(self.done), nat.ptr(self.winner), nat.ptr(self.reward), nat.stream_ptr())

        def launch():
            rc = fn(*args)
            if rc:
                nat.check(rc, "tron_step_encode")
        return launch

    def encode(self, obs_format=None, out=None):
        fmt = self._fmt if obs_format is None else nat.OBS[obs_format]
        if out is None:
            out = self.obs if fmt == self._fmt else self._alloc_obs(fmt)
        with torch.cuda.device(self.device):
            nat.check(self._lib.tron_encode(self._h, fmt, nat.ptr(out), nat.stream_ptr()), "tron_encode")
        return out

    def rollout_random(self, k_steps, totals=None, nonreversing=False, per_step_launches=False, two_streams=False,
                       resident=False):
        """k_steps random-action steps with autoreset (the BASELINE synthetic rollout): persistent launches of
        up to 64 steps each, or — per_step_launches=True — one launch per step, or — two_streams=True — one launch
        per step and per half of the envs on two streams (same results every way).  resident=True (observation-is-state
        storage): inside a persistent launch the boards stay in LDS between steps instead of being re-read from the
        observation buffer — same results, a third less HBM traffic."""
        flags = ((nat.STEP_NONREVERSING if nonreversing else 0) | (nat.ROLLOUT_PER_STEP if per_step_launches else 0) |
                 (nat.ROLLOUT_TWO_STREAMS if two_streams else 0) | (nat.ROLLOUT_RESIDENT if resident else 0))
        with torch.cuda.device(self.device):
            nat.check(self._lib.tron_rollout_random(self._h, int(k_steps), flags,
                                                    self._fmt, nat.ptr(self.obs),
                                                    nat.ptr(totals), nat.stream_ptr()), "tron_rollout_random")

    def minimax_actions(self, player, mode="voronoi", out=None, want_values=False):
        """MinimaxPlayer(2, mode).action(game.map(), player) for every env (minimax.py:284-297):
        int8 [N] actions 0..3 = UP, RIGHT, DOWN, LEFT (-1 for finished games).  With want_values
        also the root children's values int32 [N, 4] and the searched-moves bit mask int8 [N]."""
        if out is None:
            out = torch.empty(self.N, dtype=torch.int8, device=self.device)
        values = torch.empty(self.N, 4, dtype=torch.int32, device=self.device) if want_values else None
        expanded = torch.empty(self.N, dtype=torch.int8, device=self.device) if want_values else None
        with torch.cuda.device(self.device):
            nat.check(self._lib.tron_minimax_actions(self._h, int(player), 2, nat.MINIMAX[mode], nat.ptr(out),
                                                     nat.ptr(values), nat.ptr(expanded), nat.stream_ptr()),
                      "tron_minimax_actions")
        return (out, values, expanded) if want_values else out

    # -- read-back ---------------------------------------------------------
    def grid(self):
        """int8 [N, W+2, W+2] Tile values (map.py:9-17)."""
        out = torch.empty(self.N, self.S, self.S, dtype=torch.int8, device=self.device)
        with torch.cuda.device(self.device):
            nat.check(self._lib.tron_get_grid(self._h, nat.ptr(out), nat.stream_ptr()), "tron_get_grid")
        return out

    def state(self):
        dev, N = self.device, self.N
        out = dict(pos=torch.empty(N, 4, dtype=torch.int8, device=dev),
                   alive=torch.empty(N, 2, dtype=torch.int8, device=dev),
                   dir=torch.empty(N, 2, dtype=torch.int8, device=dev),
                   done=torch.empty(N, dtype=torch.int8, device=dev),
                   winner=torch.empty(N, dtype=torch.int8, device=dev),
                   weight=torch.empty(N, 2, dtype=torch.int16, device=dev),
                   degree=torch.empty(N, dtype=torch.int16, device=dev),
                   slide=torch.empty(N, dtype=torch.float64, device=dev),
                   counters=torch.empty(N, 3, dtype=torch.int32, device=dev))
        with torch.cuda.device(self.device):
            nat.check(self._lib.tron_get_state(self._h, *[nat.ptr(out[k]) for k in
                                                          ("pos", "alive", "dir", "done", "winner", "weight",
                                                           "degree", "slide", "counters")],
                                               nat.stream_ptr()), "tron_get_state")
        return out


def encode_codes(tiles, player):
    """Map.state_for_player(player) on a stack of raw tile images (map.py:67-84)."""
    t = tiles.contiguous()
    out = torch.empty_like(t)
    n = t.shape[0] if t.dim() > 2 else 1
    nat.check(nat.lib().tron_encode_codes(nat.ptr(t), n, t.numel() // n, int(player), nat.ptr(out),
                                          nat.stream_ptr()), "tron_encode_codes")
    return out


def minimax_codes(codes, draws=None, mode="voronoi", depth=2):
    """The reference's MinimaxPlayer(depth, mode) search (minimax.py:216-297) on a stack of
    observation-code images int8 [n, S, S] of the player to move.  Returns (actions int8 [n] in
    0..3 = UP, RIGHT, DOWN, LEFT, root values int32 [n, 4], searched-moves bit mask int8 [n]).
    draws: uint32-valued int64/int32 tensor [n] for random.choice / randint (None = zeros)."""
    c = codes.contiguous()
    n, side = c.shape[0], c.shape[-1]
    act = torch.empty(n, dtype=torch.int8, device=c.device)
    values = torch.empty(n, 4, dtype=torch.int32, device=c.device)
    expanded = torch.empty(n, dtype=torch.int8, device=c.device)
    d = None
    if draws is not None:
        d = draws.to(torch.int64) & 0xFFFFFFFF
        d = torch.where(d >= 2 ** 31, d - 2 ** 32, d).to(torch.int32).contiguous()      # same 32 bits
    with torch.cuda.device(c.device):
        nat.check(nat.lib().tron_minimax_codes(nat.ptr(c), n, side, int(depth), nat.MINIMAX[mode], nat.ptr(d),
                                               nat.ptr(act), nat.ptr(values), nat.ptr(expanded), nat.stream_ptr()),
                  "tron_minimax_codes")
    return act, values, expanded


def pop_up_planes(codes):
    """util.pop_up on a stack of code planes [n, S, S] -> f32 [n, 3, S, S] (util.py:11-37)."""
    c = codes.contiguous()
    n, cells = c.shape[0], c[0].numel()
    out = torch.empty((n, 3) + tuple(c.shape[1:]), dtype=torch.float32, device=c.device)
    nat.check(nat.lib().tron_pop_up(nat.ptr(c), n, cells, nat.ptr(out), nat.stream_ptr()), "tron_pop_up")
    return out


class DeviceReplay:
    """HBM ring of transitions — DDQN.ReplayBuffer (DDQN.py:167-203) without the host."""

    def __init__(self, capacity, cells, seed=0x5EED, rank=0, device=None):
        if not torch.cuda.is_available():
            raise nat.TronNativeError("DeviceReplay needs a HIP device (no CPU fallback)")
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self.capacity, self.cells = int(capacity), int(cells)
        self._lib = nat.lib()
        h = C.c_void_p()
        with torch.cuda.device(self.device):
            nat.check(self._lib.tron_replay_create(self.capacity, self.cells, seed & 0xFFFFFFFF, rank & 0xFFFFFFFF,
                                                   C.byref(h)), "tron_replay_create")
        self._h = h

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self._lib.tron_replay_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __len__(self):
        size = C.c_int64()
        nat.check(self._lib.tron_replay_size(self._h, C.byref(size), None))
        return size.value

    def add(self, state, action, reward, next_state, done):
        """Batched ReplayBuffer.add: state/next_state int8 [n, cells...] codes, action int8 [n],
        reward f32 [n], done int8 [n]."""
        n = next_state.shape[0]
        # every converted tensor stays bound to a local until the call returns: a temporary freed
        # right after nat.ptr() could be handed to the next conversion by the caching allocator
        # before the push kernel has read it
        s2 = next_state.reshape(n, -1).to(torch.int8).contiguous()
        s = s2 if state is None else state.reshape(n, -1).to(torch.int8).contiguous()     # None: add_states() wrote them
        a = action.reshape(n).to(torch.int8).contiguous()
        r = reward.reshape(n).to(torch.float32).contiguous()
        d = done.reshape(n).to(torch.int8).contiguous()
        if s.shape[1] != self.cells or s2.shape[1] != self.cells:
            raise ValueError(f"states must have {self.cells} cells per row, got {s.shape[1]} / {s2.shape[1]}")
        with torch.cuda.device(self.device):
            nat.check(self._lib.tron_replay_push(self._h, n, None if state is None else nat.ptr(s), nat.ptr(a), nat.ptr(r),
                                                 nat.ptr(s2), nat.ptr(d), nat.stream_ptr()), "tron_replay_push")
        del s, s2, a, r, d

    def add_states(self, state):
        """The `state` rows of the next add(), written ahead of it (tron_replay_push_states): call before the env step
        overwrites an observation buffer that is the env state, then add(None, action, reward, next_state, done)."""
        n = state.shape[0]
        s = state.reshape(n, -1)
        if s.dtype != torch.int8 or not s.is_contiguous() or s.shape[1] != self.cells:
            raise ValueError(f"add_states takes contiguous int8 rows of {self.cells} cells")
        with torch.cuda.device(self.device):
            nat.check(self._lib.tron_replay_push_states(self._h, n, nat.ptr(s), nat.stream_ptr()), "tron_replay_push_states")

    def sample(self, batch, channels=3, plane4=0.0, side=None):
        """ReplayBuffer.sample(): (states, actions, rewards, next_states, dones) on the device,
        states f32 [batch, channels, S, S], actions i64 [batch,1], rewards/dones f32 [batch,1]."""
        dev = self.device
        S = side if side is not None else int(round(self.cells ** 0.5))
        st = torch.empty(batch, channels, S, S, dtype=torch.float32, device=dev)
        s2 = torch.empty_like(st)
        a = torch.empty(batch, 1, dtype=torch.int64, device=dev)
        r = torch.empty(batch, 1, dtype=torch.float32, device=dev)
        d = torch.empty(batch, 1, dtype=torch.float32, device=dev)
        with torch.cuda.device(self.device):
            nat.check(self._lib.tron_replay_sample(self._h, batch, channels, float(plane4), nat.ptr(st), nat.ptr(a),
                                                   nat.ptr(r), nat.ptr(s2), nat.ptr(d), nat.stream_ptr()),
                      "tron_replay_sample")
        return st, a, r, s2, d

    def sample_codes(self, batch, side=None):
        """The same draw with the states left as int8 observation codes [batch, S, S] (tron_replay_sample_codes): what
        the learner's conv1 and the weight-stationary target forwards read — a twelfth of the bytes of the f32 planes."""
        dev = self.device
        S = side if side is not None else int(round(self.cells ** 0.5))
        st = torch.empty(batch, S, S, dtype=torch.int8, device=dev)
        s2 = torch.empty_like(st)
        a = torch.empty(batch, 1, dtype=torch.int64, device=dev)
        r = torch.empty(batch, 1, dtype=torch.float32, device=dev)
        d = torch.empty(batch, 1, dtype=torch.float32, device=dev)
        with torch.cuda.device(self.device):
            nat.check(self._lib.tron_replay_sample_codes(self._h, batch, nat.ptr(st), nat.ptr(a), nat.ptr(r), nat.ptr(s2),
                                                         nat.ptr(d), nat.stream_ptr()), "tron_replay_sample_codes")
        return st, a, r, s2, d

    def cursor(self):
        """(write head, filled slots, sample() calls so far): tron_replay_get_cursor."""
        head, size, calls = C.c_int64(), C.c_int64(), C.c_uint32()
        nat.check(self._lib.tron_replay_get_cursor(self._h, C.byref(head), C.byref(size), C.byref(calls)), "tron_replay_get_cursor")
        return head.value, size.value, calls.value

    def state_dict(self, contents=True):
        """The ring for a checkpoint ("replay head"): the cursor always; with contents=True also the filled
        slots of all five arrays as HOST tensors (2 * cells + 6 bytes per slot: 1.35 GB for 1 M slots at 24x24 boards) —
        slots [0, size), in ring order, so that load_state_dict puts every transition back where it was."""
        head, size, calls = self.cursor()
        out = {"capacity": self.capacity, "cells": self.cells, "head": head, "size": size, "sample_calls": calls}
        if contents and size > 0:
            dev = self.device
            s = torch.empty(size, self.cells, dtype=torch.int8, device=dev)
            s2 = torch.empty_like(s)
            a = torch.empty(size, dtype=torch.int8, device=dev)
            r = torch.empty(size, dtype=torch.float32, device=dev)
            d = torch.empty(size, dtype=torch.int8, device=dev)
            with torch.cuda.device(dev):
                nat.check(self._lib.tron_replay_export(self._h, 0, size, nat.ptr(s), nat.ptr(s2), nat.ptr(a), nat.ptr(r), nat.ptr(d),
                                                       nat.stream_ptr()), "tron_replay_export")
            out.update(states=s.cpu(), next_states=s2.cpu(), actions=a.cpu(), rewards=r.cpu(), dones=d.cpu())
        return out

    def load_state_dict(self, sd):
        """Inverse of state_dict(): the cursor, and the contents when the checkpoint holds them (a cursor-only checkpoint
        into a ring that has fewer slots filled than it claims is refused: sampling would read slots nobody wrote)."""
        if int(sd["capacity"]) != self.capacity or int(sd["cells"]) != self.cells:
            raise ValueError(f"checkpointed ring is {sd['capacity']} x {sd['cells']}, this one {self.capacity} x {self.cells}")
        size = int(sd["size"])
        if "states" in sd:
            dev = self.device
            t = [sd[k].to(dev).contiguous() for k in ("states", "next_states", "actions", "rewards", "dones")]
            if t[0].shape != (size, self.cells) or t[0].dtype != torch.int8 or t[3].dtype != torch.float32:
                raise ValueError("checkpointed ring contents do not match its cursor")
            with torch.cuda.device(dev):
                nat.check(self._lib.tron_replay_import(self._h, 0, size, *[nat.ptr(x) for x in t], nat.stream_ptr()), "tron_replay_import")
                torch.cuda.current_stream().synchronize()              # (t is freed on return)
        elif size > len(self):
            raise ValueError("cursor-only replay checkpoint: the ring holds fewer transitions than the cursor says")
        nat.check(self._lib.tron_replay_set_cursor(self._h, int(sd["head"]), size, int(sd["sample_calls"])), "tron_replay_set_cursor")

    def last_indices(self, batch):
        out = torch.empty(batch, dtype=torch.int64, device=self.device)
        with torch.cuda.device(self.device):
            nat.check(self._lib.tron_replay_indices(self._h, batch, nat.ptr(out), nat.stream_ptr()))
        return out
