#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by RUNNING THE REFERENCE.

Runs only in the build container (the reference lives at /root/reference and
never travels to the GPU box).  Nothing from the reference is copied: this
script imports it, drives it with recorded inputs, and stores inputs + outputs
as small .npz files.  The tests then check oracle/ (the C restatement) and the
HIP path against those files.

Import recipe: tron.game pulls in two third-party modules that
the hot path never uses (`orderedset`, `torchvision.models`); both are absent
here and are satisfied with inert stub modules.

Usage:  python tests/golden/make_golden.py            (writes next to itself)
"""
import os
import sys
import types
import random as pyrandom

_ORIG_RANDOM = pyrandom.random
_ORIG_RANDINT = pyrandom.randint

import numpy as np

REF = "/root/reference/Deep-Q-learning_TRON"
OUT = os.path.dirname(os.path.abspath(__file__))

sys.dont_write_bytecode = True
sys.path.insert(0, REF)

_os = types.ModuleType("orderedset")


class _OrderedSet(list):  # only used by dead code paths (SetQueue, minimax)
    def add(self, x):
        if x not in self:
            self.append(x)


_os.OrderedSet = _OrderedSet
sys.modules["orderedset"] = _os
_tv = types.ModuleType("torchvision")
_tvm = types.ModuleType("torchvision.models")
_tv.models = _tvm
sys.modules["torchvision"] = _tv
sys.modules["torchvision.models"] = _tvm

import tron.game as RG  # noqa: E402
import tron.util as RU  # noqa: E402
import tron.map as RM  # noqa: E402
from tron.player import ACPlayer  # noqa: E402

TILE_VALUE = np.vectorize(lambda t: t.value, otypes=[np.int8])


def raw_grid(game):
    """int8 Tile.value image of the newest map, storage order [row+1][col+1]."""
    return TILE_VALUE(game.history[-1].map._data)


def new_game(W, starts, mode=None, slide_pram=None, weight=None, degree=None):
    g = RG.Game(W, W, [RG.PositionPlayer(1, ACPlayer(), [int(starts[0]), int(starts[1])]),
                       RG.PositionPlayer(2, ACPlayer(), [int(starts[2]), int(starts[3])])],
                mode, slide_pram)
    if weight is not None:
        g.weight = [int(weight[0]), int(weight[1])]
    if degree is not None:
        g.degree = int(degree)
    return g


def snapshot(g):
    pos = [g.pps[0].position[0], g.pps[0].position[1], g.pps[1].position[0], g.pps[1].position[1]]
    alive = [int(g.pps[0].alive), int(g.pps[1].alive)]
    return pos, alive


# --------------------------------------------------------------------------
# G-step-exhaustive: every start pair x every action pair, one step, 4x4 board
# --------------------------------------------------------------------------
def gen_step_exhaustive(W=4):
    starts, actions, grids, poss, alives, dones, winners, obs1, obs2 = ([] for _ in range(9))
    cells = [(r, c) for r in range(W) for c in range(W)]
    for a in cells:
        for b in cells:
            if a == b:
                continue
            for a1 in range(4):
                for a2 in range(4):
                    g = new_game(W, (a[0], a[1], b[0], b[1]))
                    n1, n2, done = g.step(a1, a2)
                    pos, alive = snapshot(g)
                    starts.append([a[0], a[1], b[0], b[1]])
                    actions.append([a1, a2])
                    grids.append(raw_grid(g))
                    poss.append(pos)
                    alives.append(alive)
                    dones.append(int(done))
                    winners.append(0 if g.winner is None else int(g.winner))
                    obs1.append(np.asarray(n1, dtype=np.int8))
                    obs2.append(np.asarray(n2, dtype=np.int8))
    np.savez_compressed(
        os.path.join(OUT, "step_exhaustive_4x4.npz"),
        W=np.int32(W), starts=np.array(starts, np.int8), actions=np.array(actions, np.int8),
        grid=np.array(grids, np.int8), pos=np.array(poss, np.int8), alive=np.array(alives, np.int8),
        done=np.array(dones, np.int8), winner=np.array(winners, np.int8),
        obs1=np.array(obs1, np.int8), obs2=np.array(obs2, np.int8))
    print("step_exhaustive_4x4:", len(starts), "cases")


# --------------------------------------------------------------------------
# Episodes (mode None / ice / temper) with recorded actions and uniforms
# --------------------------------------------------------------------------
class UniformFeeder:
    """Replaces tron.game.random.random.  The reference draws one uniform per
    player per step, and only when that player's first target cell is in-bounds
    and EMPTY (game.py:163-169).  We hand it the value recorded for that
    (step, player) slot; the player index is the `id` local of next_frame."""

    def __init__(self):
        self.slots = None
        self.consumed = None

    def arm(self, u2):
        self.slots = u2
        self.consumed = [0, 0]

    def __call__(self):
        pid = sys._getframe(1).f_locals["id"]
        self.consumed[pid] += 1
        return float(self.slots[pid])


def safe_actions(g, W, rng, p_safe):
    """Action pair from a 'mostly avoid obstacles' policy so episodes get long."""
    m = g.history[-1].map
    out = []
    for pp in g.pps:
        if rng.random() < p_safe:
            ok = []
            for a, (dr, dc) in enumerate(((-1, 0), (0, 1), (1, 0), (0, -1))):
                r, c = pp.position[0] + dr, pp.position[1] + dc
                if 0 <= r < W and 0 <= c < W and m[r, c] is RM.Tile.EMPTY:
                    ok.append(a)
            out.append(rng.choice(ok) if ok else rng.randrange(4))
        else:
            out.append(rng.randrange(4))
    return out


def gen_episodes(name, W, n_eps, seed, mode=None, p_safe=0.8, keep_steps=False, slide_choices=(None,)):
    rng = pyrandom.Random(seed)
    feeder = UniformFeeder()
    RG.random.random = feeder  # tron.game's `random` module object (same as stdlib's)
    ep_off = [0]
    starts, slides, weights, degrees, finals, winners, fobs1, fobs2 = ([] for _ in range(8))
    acts, unis, cons, poss, alives, dones = ([] for _ in range(6))
    step_grids, step_obs1, step_obs2 = [], [], []
    try:
        for _ in range(n_eps):
            while True:
                s = [rng.randrange(W) for _ in range(4)]
                if (s[0], s[1]) != (s[2], s[3]):
                    break
            sp = rng.choice(slide_choices)
            w = [rng.randint(40, 101), rng.randint(40, 101)]
            d = rng.randint(-30, 30)
            g = new_game(W, s, mode, sp, w, d)
            starts.append(s)
            slides.append(float(g.slide))
            weights.append(w)
            degrees.append(d)
            done = False
            while not done:
                a = safe_actions(g, W, rng, p_safe)
                # f32-exact uniforms; now and then sit exactly on / next to the rate
                u = [rng.randrange(1 << 24) / float(1 << 24) for _ in range(2)]
                if mode is not None and rng.random() < 0.05:
                    u[rng.randrange(2)] = float(np.float32(g.slide))
                feeder.arm(u)
                n1, n2, done = g.step(a[0], a[1])
                pos, alive = snapshot(g)
                acts.append(a)
                unis.append(u)
                cons.append(list(feeder.consumed))
                poss.append(pos)
                alives.append(alive)
                dones.append(int(done))
                if keep_steps:
                    step_grids.append(raw_grid(g))
                    step_obs1.append(np.asarray(n1, np.int8))
                    step_obs2.append(np.asarray(n2, np.int8))
            ep_off.append(len(acts))
            finals.append(raw_grid(g))
            winners.append(0 if g.winner is None else int(g.winner))
            fobs1.append(np.asarray(g.next_p1, np.int8))
            fobs2.append(np.asarray(g.next_p2, np.int8))
    finally:
        RG.random.random = _ORIG_RANDOM
    d = dict(W=np.int32(W), mode=np.array(mode if mode else "none"),
             ep_off=np.array(ep_off, np.int32), starts=np.array(starts, np.int8),
             slide=np.array(slides, np.float64), weight=np.array(weights, np.int16),
             degree=np.array(degrees, np.int16),
             actions=np.array(acts, np.int8), uniforms=np.array(unis, np.float32),
             consumed=np.array(cons, np.int8), pos=np.array(poss, np.int8),
             alive=np.array(alives, np.int8), done=np.array(dones, np.int8),
             winner=np.array(winners, np.int8), final_grid=np.array(finals, np.int8),
             final_obs1=np.array(fobs1, np.int8), final_obs2=np.array(fobs2, np.int8))
    if keep_steps:
        d.update(step_grid=np.array(step_grids, np.int8), step_obs1=np.array(step_obs1, np.int8),
                 step_obs2=np.array(step_obs2, np.int8))
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **d)
    lens = np.diff(ep_off)
    print(f"{name}: {n_eps} episodes, {len(acts)} steps, mean len {lens.mean():.1f}, max {lens.max()},"
          f" winners {np.bincount(np.array(winners), minlength=3).tolist()},"
          f" uniforms consumed {int(np.array(cons).sum())}")


# --------------------------------------------------------------------------
# G-encode: arbitrary tile images -> state_for_player -> pop_up (+ prob plane)
# --------------------------------------------------------------------------
def gen_encode():
    rng = np.random.RandomState(7)
    tiles = {t.value: t for t in RM.Tile}
    out = {}
    for W, n in ((4, 64), (10, 32), (24, 6)):
        S = W + 2
        raw = rng.randint(-1, 7, size=(n, S, S)).astype(np.int8)
        codes = np.zeros((n, 2, S, S), np.int8)
        planes = np.zeros((n, 2, 3, S, S), np.float64)
        for k in range(n):
            m = RM.Map(W, W, RM.Tile.EMPTY, RM.Tile.WALL)
            m._data = np.array([[tiles[int(v)] for v in row] for row in raw[k]])
            for p in (1, 2):
                c = m.state_for_player(p)
                codes[k, p - 1] = c
                planes[k, p - 1] = RU.pop_up(c)
        out[f"raw_{W}"] = raw
        out[f"codes_{W}"] = codes
        out[f"planes_{W}"] = planes
    # env scalars: Game.get_rate / get_degree_silde / get_multy / prob_map (game.py:96-147)
    g = new_game(10, (0, 0, 5, 5))
    degs = np.arange(-30, 31)
    wts = np.arange(40, 102)
    rate = np.zeros((len(degs), len(wts)), np.float64)
    rate_none = np.zeros(len(degs), np.float64)
    for i, dg in enumerate(degs):
        g.degree = int(dg)
        rate_none[i] = g.get_rate()
        for j, wt in enumerate(wts):
            g.weight = [int(wt), 40]
            rate[i, j] = g.get_rate(0)
    slides = np.array([0.0, 0.03, 0.06, 0.09, 0.12, 0.15, 0.18, 0.21, 0.24, 0.27, 0.3, 0.33, 0.36, 0.25, 0.5])
    dslide = np.zeros(len(slides), np.float64)
    for i, s in enumerate(slides):
        g.slide = float(s)
        dslide[i] = g.get_degree_silde()
    g.slide = 0.15
    g.degree = -7
    g.weight = [55, 99]
    out.update(rate_degrees=degs.astype(np.int16), rate_weights=wts.astype(np.int16), rate=rate,
               rate_none=rate_none, slides=slides, degree_slide=dslide,
               prob_map_015=np.asarray(g.prob_map(), np.float64),
               degree_map_m7=np.asarray(g.degree_map(), np.float64),
               multy0=np.array(g.get_multy(0), np.float64), multy1=np.array(g.get_multy(1), np.float64),
               util_prob_map_3=np.asarray(RU.prob_map(3.0), np.float64))
    np.savez_compressed(os.path.join(OUT, "encode.npz"), **out)
    print("encode: done")


# --------------------------------------------------------------------------
# G-reset: make_game / Game.__init__ under a replayed randint stream
# --------------------------------------------------------------------------
class RandintFeeder:
    """random.randint(a, b) := a + ((u32 * (b - a + 1)) >> 32) over a recorded
    u32 stream, so the order AND the ranges of the reference's draws are pinned
    (util.py:48-78, game.py:83,87)."""

    def __init__(self, stream):
        self.stream = [int(x) for x in stream]
        self.i = 0
        self.calls = []

    def __call__(self, a, b):
        u = self.stream[self.i]
        self.i += 1
        self.calls.append((a, b))
        return a + ((u * (b - a + 1)) >> 32)


def gen_reset():
    rng = np.random.RandomState(11)
    out = {}
    for W in (4, 10, 24):
        RU.MAP_WIDTH = RU.MAP_HEIGHT = W
        for mode in (None, "fair"):
            n = 400 if W == 4 else 150
            streams = rng.randint(0, 1 << 32, size=(n, 48), dtype=np.uint64).astype(np.uint32)
            if W == 4 and mode is None:
                # force clashes: x2,y2 equal to x1,y1 for several redraw rounds
                for k in range(0, 60):
                    streams[k, 2] = streams[k, 0]
                    streams[k, 3] = streams[k, 1]
                for k in range(0, 20):
                    streams[k, 4] = streams[k, 0]
                    streams[k, 5] = streams[k, 1]
            res = np.zeros((n, 8), np.int16)  # x1 y1 x2 y2 w0 w1 degree ndraws
            grids = np.zeros((n, W + 2, W + 2), np.int8)
            for k in range(n):
                f = RandintFeeder(streams[k])
                RU.random.randint = f
                try:
                    g = RU.make_game(True, True, mode=mode, gamemode="temper")
                finally:
                    RU.random.randint = _ORIG_RANDINT
                p1, p2 = g.pps[0].position, g.pps[1].position
                res[k] = [p1[0], p1[1], p2[0], p2[1], g.weight[0], g.weight[1], g.degree, f.i]
                grids[k] = raw_grid(g)
            tag = f"{W}_{mode or 'default'}"
            out["stream_" + tag] = streams
            out["result_" + tag] = res
            out["grid_" + tag] = grids
    RU.MAP_WIDTH = RU.MAP_HEIGHT = 10
    np.savez_compressed(os.path.join(OUT, "reset.npz"), **out)
    print("reset: done")


# --------------------------------------------------------------------------
# G-reward: get_reward (util.py:87-94) + the trainers' literal tables
# --------------------------------------------------------------------------
def gen_reward():
    import config as RC
    rows = []
    g = new_game(4, (0, 0, 3, 3))
    for ci, cons in enumerate((RC.reward_cons1, RC.reward_cons2, RC.reward_cons3)):
        for w in (None, 1, 2):
            g.winner = w
            r = RU.get_reward(g, cons)
            rows.append([ci, 0 if w is None else w, float(cons[0]), float(cons[1]), float(r[0]), float(r[1])])
    np.savez_compressed(os.path.join(OUT, "reward.npz"), get_reward=np.array(rows, np.float64))
    print("reward: done")


# --------------------------------------------------------------------------
# G-net: Q-values of the reference DQNNet.Net and one full DDQN Agent.learn() step
# --------------------------------------------------------------------------
def gen_net():
    import collections
    import json
    import torch
    sys.path.insert(0, OUT)
    from netgen import det_state_dict
    # torch.utils.tensorboard is imported by DDQN.py only for logging; tensorboard is absent here
    tb = types.ModuleType("torch.utils.tensorboard")
    tb.SummaryWriter = type("SummaryWriter", (), {"__init__": lambda self, *a, **k: None,
                                                  "add_scalar": lambda self, *a, **k: None})
    sys.modules["torch.utils.tensorboard"] = tb
    import Net.DQNNet as RD
    import Net.ACNet as RA
    RD.Net.mish = RA.Net.mish                 # DQNNet.Net never defines mish (DQNNet.py:31)
    import DDQN as RDD

    torch.manual_seed(0)
    net = RD.Net()
    shapes = collections.OrderedDict((k, tuple(v.shape)) for k, v in net.state_dict().items())
    net.load_state_dict(det_state_dict(shapes, salt=0))
    net.eval()
    enc = np.load(os.path.join(OUT, "encode.npz"))
    planes = enc["planes_10"].astype(np.float32)                  # [32, 2, 3, 12, 12]
    x3 = planes[:8, 0]
    x = np.concatenate([x3, np.full((8, 1, 12, 12), 5.0, np.float32)], axis=1)    # + prob_map plane (slide .15)
    with torch.no_grad():
        q = net(torch.from_numpy(x)).numpy()

    # one real Agent.learn() with dropout switched off (p = 0) so train mode is deterministic
    agent = RDD.Agent()
    agent.qnetwork_local.load_state_dict(det_state_dict(shapes, salt=1))
    agent.qnetwork_target.load_state_dict(det_state_dict(shapes, salt=2))
    agent.qnetwork_local.dropout.p = 0.0
    agent.qnetwork_target.dropout.p = 0.0
    B = 16
    s = np.concatenate([planes[:B, 0], np.full((B, 1, 12, 12), 5.0, np.float32)], axis=1)
    s2 = np.concatenate([planes[B:2 * B, 1], np.full((B, 1, 12, 12), 5.0, np.float32)], axis=1)
    rs = np.random.RandomState(3)
    a = rs.randint(0, 4, size=(B, 1)).astype(np.int64)
    r = rs.choice([-1.0, 100.0, -100.0, 0.0], size=(B, 1)).astype(np.float32)
    d = (rs.rand(B, 1) < 0.3).astype(np.float32)
    exp = tuple(torch.from_numpy(t) for t in (s, a, r, s2, d))
    with torch.no_grad():
        q_local_before = agent.qnetwork_local.eval()(exp[0]).numpy()
    agent.learn(exp, RDD.GAMMA)
    loss = float(agent.totalloss)
    after_local = {k: v.detach().numpy().copy() for k, v in agent.qnetwork_local.state_dict().items()}
    after_target = {k: v.detach().numpy().copy() for k, v in agent.qnetwork_target.state_dict().items()}
    probe = ["conv1.weight", "conv4.bias", "conv7.weight", "fc1.weight", "actor2.weight", "actor2.bias"]
    out = dict(shapes_json=np.array(json.dumps({k: list(v) for k, v in shapes.items()})),
               x=x, q=q, ls=s, la=a, lr=r, ls2=s2, ld=d, loss=np.float64(loss), gamma=np.float64(RDD.GAMMA),
               tau=np.float64(RDD.TAU), q_local_before=q_local_before)
    for k in probe:
        out["local_" + k] = after_local[k].reshape(-1)[:512]
        out["target_" + k] = after_target[k].reshape(-1)[:512]
    np.savez_compressed(os.path.join(OUT, "net.npz"), **out)
    print("net: Q", q.shape, "loss", loss, "params", sum(int(np.prod(v)) for v in shapes.values()))


# --------------------------------------------------------------------------
# G-acktr: actor-critic nets, RolloutStorage returns, A2C and ACKTR (KFAC) updates
# --------------------------------------------------------------------------
def gen_acktr():
    """Harness patches (the reference's ACKTR path assumes a GPU and an old torch):
    Tensor.cuda() is the identity here (ACNet.py:94,164,233,300 call it unconditionally),
    torch.symeig (removed in torch 2) is served by torch.linalg.eigh (kfac.py:220-223),
    torch.utils.tensorboard is a stub.  Nothing else is changed."""
    import collections
    import json
    import warnings
    import torch
    warnings.filterwarnings("ignore")
    sys.path.insert(0, OUT)
    from netgen import det_state_dict
    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.symeig = lambda A, eigenvectors=True: torch.linalg.eigh(A)
    tb = types.ModuleType("torch.utils.tensorboard")
    tb.SummaryWriter = type("SummaryWriter", (), {"__init__": lambda self, *a, **k: None,
                                                  "add_scalar": lambda self, *a, **k: None})
    sys.modules["torch.utils.tensorboard"] = tb
    import Net.ACNet as RA
    import ACKTR as RK

    enc = np.load(os.path.join(OUT, "encode.npz"))
    planes = enc["planes_10"].astype(np.float32)                  # [32, 2, 3, 12, 12]
    rs = np.random.RandomState(17)
    out = {}
    B = 6
    x3 = planes[:B, 0]
    x4 = np.concatenate([x3, np.full((B, 1, 12, 12), 5.0, np.float32)], 1)
    env1 = rs.uniform(-0.3, 0.6, B).astype(np.float32)            # a rate-like scalar
    env2 = np.stack([rs.randint(-30, 31, B), rs.randint(40, 102, B)], 1).astype(np.float32)   # get_multy
    out.update(x3=x3, x4=x4, env1=env1, env2=env2)
    shapes_all = {}
    for name, cls, args in (("MapNet", RA.MapNet, (x4,)), ("TestNet", RA.TestNet, (x3, env1)),
                            ("Net3", RA.Net3, (x3, env1)), ("Net4", RA.Net4, (x3, env1)),
                            ("Mulnet", RA.Mulnet, (x3, env2))):
        net = cls()
        shapes = collections.OrderedDict((k, tuple(v.shape)) for k, v in net.state_dict().items())
        shapes_all[name] = {k: list(v) for k, v in shapes.items()}
        net.load_state_dict(det_state_dict(shapes, salt=3))
        net.eval()
        with torch.no_grad():
            value, logits = net(*[torch.from_numpy(a) for a in args])
            acts = torch.from_numpy(rs.randint(0, 4, (B, 1)).astype(np.int64))
            v2, lp, ent = net.evaluate_actions(*([torch.from_numpy(args[0]), acts] +
                                                 [torch.from_numpy(a) for a in args[1:]]))
        out[name + "_value"] = value.numpy()
        out[name + "_logits"] = logits.numpy()
        out[name + "_acts"] = acts.numpy()
        out[name + "_logp"] = lp.numpy()
        out[name + "_entropy"] = np.float64(float(ent))
    out["shapes_json"] = np.array(json.dumps(shapes_all))

    # rollouts + updates, reference globals: 16 processes x 5 steps
    T, N = RK.NUM_ADVANCED_STEP, RK.NUM_PROCESSES
    idx = rs.randint(0, 32, size=(T + 1, N))
    pl = rs.randint(0, 2, size=(T + 1, N))
    obs3 = planes[idx, pl]                                        # [T+1, N, 3, 12, 12]
    obs4 = np.concatenate([obs3, np.full((T + 1, N, 1, 12, 12), 5.0, np.float32)], 2)
    actions = rs.randint(0, 4, size=(T, N, 1)).astype(np.int64)
    rewards = rs.choice([-1.0, 20.0, -10.0, 0.0], size=(T, N, 1), p=[.7, .1, .1, .1]).astype(np.float32)
    masks = (rs.rand(T + 1, N, 1) > 0.25).astype(np.float32)
    probs = np.stack([rs.randint(-30, 31, (T, N)), rs.randint(40, 102, (T, N))], 2).astype(np.float32)
    next_value = rs.randn(N, 1).astype(np.float32)
    out.update(r_obs3=obs3, r_actions=actions, r_rewards=rewards, r_masks=masks, r_probs=probs, r_next=next_value)
    probe = ["conv1.module.weight", "conv1.add_bias._bias", "conv7.module.weight", "fc1.module.weight",
             "actor2.module.weight", "critic3.add_bias._bias"]

    def fill(ro, obs, with_probs):
        ro.observations.copy_(torch.from_numpy(obs))
        ro.actions.copy_(torch.from_numpy(actions))
        ro.rewards.copy_(torch.from_numpy(rewards))
        ro.masks.copy_(torch.from_numpy(masks))
        if with_probs:
            ro.probs.copy_(torch.from_numpy(probs))
        ro.compute_returns(torch.from_numpy(next_value))
        return ro

    args_ns = types.SimpleNamespace(p=None, v=None)
    for tag, cls, obs, with_probs in (("map", RA.MapNet, obs4, False), ("mul", RA.Mulnet, obs3, True)):
        RK.Nettype = cls if with_probs else RK.maptype            # the module global Brain.update branches on
        for mode in ("a2c", "acktr"):
            net = cls()
            brain = RK.Brain(net, args_ns, acktr=(mode == "acktr"))
            shapes = collections.OrderedDict((k, tuple(v.shape)) for k, v in net.state_dict().items())
            net.load_state_dict(det_state_dict(shapes, salt=4))
            net.dropout.p = 0.0                                   # deterministic train mode
            ro = fill(RK.RolloutStorage(T, N), obs, with_probs)
            if tag == "map" and mode == "a2c":
                out["returns"] = ro.returns.numpy().copy()
            for k in range(2 if mode == "acktr" else 1):
                torch.manual_seed(1000 + k)                       # value_noise of the sampled Fisher
                res = brain.update(ro)
                out[f"{tag}_{mode}_stats{k}"] = np.array([float(t) for t in res], np.float64)
                sd = net.state_dict()
                for name in probe:
                    key = name if name in sd else name.replace(".module.weight", ".weight").replace(".add_bias._bias", ".bias")
                    out[f"{tag}_{mode}_u{k}_{name}"] = sd[key].detach().numpy().reshape(-1)[:384].copy()
            if mode == "acktr":
                opt = brain.optimizer
                mods = dict(net.named_modules())
                for mn in ("conv1.module", "conv7.module", "fc1.module", "actor2.add_bias"):
                    out[f"{tag}_maa_{mn}"] = opt.m_aa[mods[mn]].numpy().reshape(-1)[:256].copy()
                    out[f"{tag}_mgg_{mn}"] = opt.m_gg[mods[mn]].numpy().reshape(-1)[:256].copy()
                out[f"{tag}_shapes_split"] = np.array(json.dumps({k: list(v) for k, v in shapes.items()}))
    np.savez_compressed(os.path.join(OUT, "acktr.npz"), **out)
    print("acktr: done;", {k: out[k].tolist() for k in out if k.endswith("stats0")})


# --------------------------------------------------------------------------
# G-minimax: MinimaxPlayer.action (tron/minimax.py) on live boards
# --------------------------------------------------------------------------
class StreamRandom:
    """Stands in for the `random` module inside tron.minimax: draws come from a recorded u32
    stream with randint(1,4) := 1 + mulhi(u,4), choice(seq) := seq[mulhi(u,len(seq))]."""

    def __init__(self, stream):
        self.s, self.i = stream, 0

    def _next(self):
        v = int(self.s[self.i])
        self.i += 1
        return v

    def randint(self, a, b):
        assert (a, b) == (1, 4)
        return 1 + ((self._next() * 4) >> 32)

    def choice(self, seq):
        return seq[(self._next() * len(seq)) >> 32]


def _board_from_tiles(W, tiles):
    """Map with interior tiles[r][c] (Tile.value ints)."""
    by_val = {t.value: t for t in RM.Tile}
    m = RM.Map(W, W, RM.Tile.EMPTY, RM.Tile.WALL)
    for r in range(W):
        for c in range(W):
            m[r, c] = by_val[int(tiles[r][c])]
    return m


def _random_board(W, rng, fill, slide_tiles):
    tiles = np.zeros((W, W), np.int8)
    for r in range(W):
        for c in range(W):
            if rng.random() < fill:
                tiles[r, c] = rng.choice((1, 3, 5, 6) if slide_tiles else (1, 3))
    cells = [(r, c) for r in range(W) for c in range(W)]
    h1 = rng.choice(cells)
    if rng.random() < 0.4:      # heads close together: crashes and contested cells
        near = [(r, c) for (r, c) in cells if (r, c) != h1 and abs(r - h1[0]) + abs(c - h1[1]) <= 2]
        h2 = rng.choice(near) if near else rng.choice([x for x in cells if x != h1])
    else:
        h2 = rng.choice([x for x in cells if x != h1])
    tiles[h1] = 2
    tiles[h2] = 4
    return tiles


def gen_minimax():
    from netgen import mm_stream
    import tron.minimax as RMM
    from tron.player import Direction
    rng = pyrandom.Random(909)
    out = {}
    summary = []
    plan = [(3, 60, (2, 4)), (4, 120, (2, 4)), (5, 80, (2, 4)), (6, 80, (2, 4)), (10, 160, (2,)), (11, 40, (2,)),
            (24, 40, (2,)), (32, 16, (2,))]
    for W, n_boards, depths in plan:
        boards = []
        # (a) random tile soups at several densities, some with slide tiles
        for k in range(n_boards // 2):
            boards.append(_random_board(W, rng, rng.choice((0.0, 0.1, 0.25, 0.4, 0.6, 0.8)), k % 5 == 0))
        # (b) positions out of 'mostly safe' self-play (long snakes, rooms)
        while len(boards) < n_boards:
            while True:
                st = [rng.randrange(W) for _ in range(4)]
                if (st[0], st[1]) != (st[2], st[3]):
                    break
            g = new_game(W, st)
            stop = rng.randrange(1, 3 * W)
            done = False
            for _ in range(stop):
                last = raw_grid(g)[1:-1, 1:-1].copy()
                a = safe_actions(g, W, rng, 0.95)
                _, _, done = g.step(a[0], a[1])
                if done:
                    break
            boards.append(last if done else raw_grid(g)[1:-1, 1:-1].copy())
        recs = {k: [] for k in ("player", "depth", "mode", "seed", "codes", "move", "values", "expanded", "draws")}
        for tiles in boards:
            m = _board_from_tiles(W, tiles)
            for pid in (1, 2):
                for depth in depths:
                    for mode_id, mode in ((0, "voronoi"), (1, RMM.Mode.DISTWALL)):
                        if depth == 4 and (rng.random() < 0.5 or (W > 5 and mode_id == 0 and rng.random() < 0.7)):
                            continue
                        seed = rng.getrandbits(32)
                        stream = mm_stream(seed, 96).astype(np.uint64)
                        sr = StreamRandom(stream)
                        RMM.random = sr
                        try:
                            mp = RMM.MinimaxPlayer(depth, mode)
                            d = mp.action(m, pid)
                        finally:
                            RMM.random = pyrandom
                        values = np.zeros(4, np.int32)
                        expanded = np.zeros(4, np.int8)
                        for ch in mp.minimax.root._children:
                            values[ch.get_action() - 1] = ch.get_value()
                            expanded[ch.get_action() - 1] = 1
                        recs["player"].append(pid)
                        recs["depth"].append(depth)
                        recs["mode"].append(mode_id)
                        recs["seed"].append(seed)
                        assert sr.i <= 96
                        recs["codes"].append(np.asarray(m.state_for_player(pid), np.int8))
                        recs["move"].append({Direction.UP: 1, Direction.RIGHT: 2, Direction.DOWN: 3, Direction.LEFT: 4}[d])
                        recs["values"].append(values)
                        recs["expanded"].append(expanded)
                        recs["draws"].append(sr.i)
        n = len(recs["move"])
        dt = dict(player=np.int8, depth=np.int8, mode=np.int8, seed=np.uint32, codes=np.int8,
                  move=np.int8, values=np.int32, expanded=np.int8, draws=np.int16)
        for k, v in recs.items():
            out[f"W{W}_{k}"] = np.array(v, dt[k])
        ex = np.array(recs["expanded"])
        summary.append(f"W={W}: {n} searches, root all-blocked {int((ex.sum(1) == 0).sum())},"
                       f" value range [{np.array(recs['values']).min()}, {np.array(recs['values']).max()}]")
    out["widths"] = np.array([p[0] for p in plan], np.int32)
    np.savez_compressed(os.path.join(OUT, "minimax.npz"), **out)
    print("minimax:", "; ".join(summary))


# --------------------------------------------------------------------------
# G-main-loop: Game.main_loop (game.py:279-328) driven by a scripted deterministic "network"
# --------------------------------------------------------------------------
def gen_main_loop():
    """Both branches of game.py:296-304: a plain model gets (planes[1,3,S,S], env scalars) — get_multy(0) for
    player 1, [get_rate()] for player 2 — and a MapNet-typed model gets the planes + the constant prob_map
    plane [1,4,S,S].  The reference tests `type(model) == maptype`, so the MapNet branch is exercised with a real
    reference MapNet whose `act` is replaced on the instance (its weights are never used).  Only player 1 can be
    a MapNet there: for model2 the branch assigns action1 (game.py:301-302) and crashes.
    Modes whose outcome does not depend on the slide uniform: None, ice with slide 1.0 (u <= 1 always) and
    ice with slide -1.0 (never)."""
    from scripted import ScriptedModel
    from Net.ACNet import MapNet
    rng = np.random.RandomState(77)
    cases = []
    for k in range(24):
        W = 10
        mode, slide = [(None, None), ("ice", 1.0), ("ice", -1.0)][k % 3]
        branch = "map" if k % 2 else "plain"
        while True:
            starts = rng.randint(0, W, 4)
            if (starts[0], starts[1]) != (starts[2], starts[3]):
                break
        weight, degree = rng.randint(40, 102, 2), int(rng.randint(-30, 31))
        # the reference sizes prob_map by config.MAP_WIDTH (=10), which is this board
        g = new_game(W, starts, mode, slide, weight, degree)
        log1, log2 = [], []
        m2 = ScriptedModel(salt=k + 1, log=log2)
        if branch == "map":
            m1 = MapNet()
            s1 = ScriptedModel(salt=k, log=log1)
            m1.act = s1.act                                  # instance attribute: type(m1) stays MapNet
        else:
            m1 = ScriptedModel(salt=k, log=log1)
        g.main_loop(m1, pop=RU.pop_up, window=None, model2=m2)
        pos, alive = snapshot(g)
        cases.append(dict(W=W, mode=mode or "none", slide=-999.0 if slide is None else slide, branch=branch,
                          starts=starts, weight=weight, degree=degree, salt=k, n_steps=len(log1),
                          winner=0 if g.winner is None else g.winner, grid=raw_grid(g), pos=pos, alive=alive,
                          history=len(g.history), log1=log1, log2=log2))
    out = {"n": len(cases)}
    for i, c in enumerate(cases):
        pre = "c%d_" % i
        out[pre + "mode"] = np.array(c["mode"])
        out[pre + "branch"] = np.array(c["branch"])
        out[pre + "scalars"] = np.array([c["W"], c["salt"], c["n_steps"], c["winner"], c["history"], c["degree"]], np.int64)
        out[pre + "slide"] = np.array(c["slide"], np.float64)
        out[pre + "starts"] = np.array(c["starts"], np.int8)
        out[pre + "weight"] = np.array(c["weight"], np.int16)
        out[pre + "grid"] = c["grid"]
        out[pre + "pos"] = np.array(c["pos"], np.int8)
        out[pre + "alive"] = np.array(c["alive"], np.int8)
        for who, log in (("p1", c["log1"]), ("p2", c["log2"])):
            out[pre + who + "_action"] = np.array([e["action"] for e in log], np.int8)
            out[pre + who + "_channels"] = np.array([e["channels"] for e in log], np.int8)
            out[pre + who + "_checksum"] = np.array([e["checksum"] for e in log], np.float64).reshape(len(log), 3)
            out[pre + who + "_plane4"] = np.array([e["plane4"] for e in log], np.float64)
            out[pre + who + "_env"] = np.array([e["env"] for e in log], np.float64).reshape(len(log), -1)
    np.savez_compressed(os.path.join(OUT, "main_loop.npz"), **out)
    print("main_loop:", len(cases), "games, steps", [c["n_steps"] for c in cases])


def main():
    if "--main-loop-only" in sys.argv:
        gen_main_loop()
        return
    if "--net-only" in sys.argv:
        gen_net()
        return
    if "--acktr-only" in sys.argv:
        gen_acktr()
        return
    if "--minimax-only" in sys.argv:
        gen_minimax()
        return
    gen_step_exhaustive()
    gen_episodes("episodes_none_4", 4, 300, 101, None, p_safe=0.6, keep_steps=True)
    gen_episodes("episodes_none_10", 10, 200, 102, None, p_safe=0.85, keep_steps=True)
    gen_episodes("episodes_none_24", 24, 40, 103, None, p_safe=0.93)
    gen_episodes("episodes_none_32", 32, 24, 104, None, p_safe=0.93)
    gen_episodes("episodes_uniform_10", 10, 400, 105, None, p_safe=0.0)
    sl = (None, 0.0, 0.03, 0.25, 0.36, 0.5, 1.0)
    gen_episodes("episodes_ice_4", 4, 300, 201, "ice", p_safe=0.6, keep_steps=True, slide_choices=sl)
    gen_episodes("episodes_ice_10", 10, 200, 202, "ice", p_safe=0.85, slide_choices=sl)
    gen_episodes("episodes_ice_24", 24, 30, 203, "ice", p_safe=0.93, slide_choices=sl)
    gen_episodes("episodes_temper_4", 4, 300, 301, "temper", p_safe=0.6, keep_steps=True)
    gen_episodes("episodes_temper_10", 10, 200, 302, "temper", p_safe=0.85)
    gen_episodes("episodes_temper_24", 24, 30, 303, "temper", p_safe=0.93)
    gen_encode()
    gen_reset()
    gen_reward()
    gen_minimax()
    gen_net()
    gen_acktr()
    gen_main_loop()


if __name__ == "__main__":
    main()
