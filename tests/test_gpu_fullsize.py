"""GPU parity at the BENCHMARKED sizes (BASELINE.json configs 3 and 5), through the C ABI:

* the persistent rollout kernels behind tron_rollout_random (k_obs_roll for mode None, k_tile_roll for the
  sliding modes) with more workgroups than the chip holds at once — late workgroups start while early ones
  are mid-rollout — against the CPU oracle stepped the same number of times on ALL envs, bit for bit;
* the recorded reference `Agent.learn()` (tests/golden/net.npz) replayed on the device;
* the replay ring at config 3's 1 M slots x 676 cells, pushed past its wrap, and the without-replacement
  sampler at every batch size.
"""
import collections
import json
import os
import sys

import numpy as np
import pytest

from conftest import GOLDEN, load_golden

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")


@pytest.fixture(scope="module")
def T():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import tron.vec as tv
    import oracle
    oracle.set_threads(min(16, len(os.sched_getaffinity(0))))
    yield tv, oracle
    oracle.set_threads(1)


def np_(t):
    return t.detach().cpu().numpy()


@pytest.mark.parametrize("N,W,mode,K", [
    (65536, 24, None, 70),        # BASELINE config 3 = the bench workload: k_obs_roll, 2 048 workgroups, 64 + 6 steps
    (65536, 24, "temper", 70),    # same size on the board-owning layout: k_tile_roll
    (16384, 32, "temper", 66),    # BASELINE config 5's env shape
    (65536, 24, "ice", 33),
])
@pytest.mark.parametrize("resident", [False, True])
def test_persistent_rollout_at_benchmarked_size(T, N, W, mode, K, resident):
    """resident=True: TRON_ROLLOUT_RESIDENT, the boards stay in LDS between the steps of a launch (mode None only;
    the flag is ignored elsewhere)."""
    tv, oracle = T
    env = tv.VecTron(N, W, mode=mode, seed=0x5EED, rank=3, obs_format="codes")
    assert env.obs_is_state                         # every mode on the observation-is-state layout (k_obs_roll / k_obs_roll_slide)
    ref = oracle.VecOracle(N, W, mode=mode, seed=0x5EED, stream=3)
    env.reset()
    ref.reset_all()
    totals = torch.zeros(4, dtype=torch.int64, device="cuda")
    env.rollout_random(K, totals, resident=resident)
    exp = np.zeros(4, np.int64)
    o = None
    for k in range(K):
        o, d, w, _ = ref.step(autoreset=True, want_obs=(k == K - 1))
        exp += [N, int(((d == 1) & (w == 1)).sum()), int(((d == 1) & (w == 2)).sum()), int(((d == 1) & (w == 0)).sum())]
    assert np.array_equal(np_(totals), exp)
    assert np.array_equal(np_(env.obs).reshape(N, 2, -1), o)
    st = {k: np_(v) for k, v in env.state().items()}
    assert np.array_equal(np_(env.grid()).reshape(N, -1), ref.grid)
    assert np.array_equal(st["pos"], ref.pos) and np.array_equal(st["alive"], ref.alive)
    assert np.array_equal(st["dir"], ref.dir)
    assert np.array_equal(st["weight"], ref.weight) and np.array_equal(st["degree"], ref.degree)
    assert np.array_equal(st["counters"][:, 0].astype(np.uint32), ref.tick)
    assert np.array_equal(st["counters"][:, 1].astype(np.uint32), ref.episode)
    assert np.array_equal(st["counters"][:, 2].astype(np.uint32), ref.eplen)
    assert int(ref.episode.max()) > 3               # games ended and restarted inside the launch


def test_learn_step_fixture_on_gpu():
    """tests/golden/net.npz's recorded DDQN Agent.learn() (DDQN.py:115-165) on cuda — through the HIP
    bias+residual+mish kernels and MIOpen: loss and probed weights of both nets within 1e-5 of the
    reference, like tests/test_net_cpu.py does on the host."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    sys.path.insert(0, GOLDEN)
    from netgen import det_state_dict
    import DDQN
    g = load_golden("net")
    shapes = collections.OrderedDict((k, tuple(v)) for k, v in json.loads(str(g["shapes_json"])).items())
    agent = DDQN.Agent(10, 4, device="cuda", make_memory=False)
    agent.qnetwork_local.load_state_dict(det_state_dict(shapes, salt=1))
    agent.qnetwork_target.load_state_dict(det_state_dict(shapes, salt=2))
    agent.qnetwork_local.dropout.p = 0.0
    agent.qnetwork_target.dropout.p = 0.0
    exp = tuple(torch.from_numpy(g[k]).cuda() for k in ("ls", "la", "lr", "ls2", "ld"))
    with torch.no_grad():
        qb = np_(agent.qnetwork_local.eval()(exp[0]))
    assert np.allclose(qb, g["q_local_before"], rtol=1e-5, atol=1e-4)
    loss = float(agent.learn(exp, DDQN.GAMMA))
    assert abs(loss - float(g["loss"])) <= 1e-5 * abs(float(g["loss"]))
    for net, tag in ((agent.qnetwork_local, "local_"), (agent.qnetwork_target, "target_")):
        sd = net.state_dict()
        for k in ("conv1.weight", "conv4.bias", "conv7.weight", "fc1.weight", "actor2.weight", "actor2.bias"):
            got = np_(sd[k]).reshape(-1)[:512]
            assert np.allclose(got, g[tag + k], rtol=1e-5, atol=1e-5), (tag, k, np.abs(got - g[tag + k]).max())


def _synthetic_rows(first, n, cells, salt):
    """Deterministic code rows for global push indices [first, first + n): a function of the index alone,
    so any slot's expected content can be rebuilt from the index of the push that last wrote it."""
    vals = torch.tensor([1, -1, -2, -3, 10, -10], dtype=torch.int8, device="cuda")
    g = torch.arange(first, first + n, device="cuda", dtype=torch.int64)[:, None]
    c = torch.arange(cells, device="cuda", dtype=torch.int64)[None, :]
    return vals[((g * 2654435761 + c * 40503 + salt * 7919) >> 7) % 6]


def test_replay_ring_at_one_million_slots(T):
    """BASELINE config 3's replay: 1 M slots x 676 cells (1.35 GB), pushed 131 072 rows at a time (one
    65 536-env step) past the wrap; sampled rows equal pop_up of the rows last pushed to those slots."""
    tv, _ = T
    cap, cells, rows = 1 << 20, 676, 131072
    rb = tv.DeviceReplay(cap, cells, seed=77)
    total = 0
    for _ in range(9):                                   # 1 179 648 pushes > capacity
        s, s2 = _synthetic_rows(total, rows, cells, 0), _synthetic_rows(total, rows, cells, 1)
        g = torch.arange(total, total + rows, device="cuda")
        rb.add(s, (g % 4).to(torch.int8), (g % 1000).to(torch.float32) - 500.0, s2, (g % 3 == 0).to(torch.int8))
        total += rows
        assert len(rb) == min(total, cap)
    for batch in (64, 4096):
        st, a, r, s2, d = rb.sample(batch, channels=3, side=26)
        idx = rb.last_indices(batch)
        assert int(idx.min()) >= 0 and int(idx.max()) < cap and idx.unique().numel() == batch
        g = idx + cap * ((total - 1 - idx) // cap)       # the last push that landed in each slot
        exp_s = torch.cat([_synthetic_rows(int(x), 1, cells, 0) for x in g[:64]])
        assert torch.equal(st[:64], tv.pop_up_planes(exp_s.reshape(-1, 26, 26)))
        exp_s2 = torch.cat([_synthetic_rows(int(x), 1, cells, 1) for x in g[:64]])
        assert torch.equal(s2[:64], tv.pop_up_planes(exp_s2.reshape(-1, 26, 26)))
        assert torch.equal(a.ravel(), g % 4) and torch.equal(r.ravel(), (g % 1000).to(torch.float32) - 500.0)
        assert torch.equal(d.ravel(), (g % 3 == 0).to(torch.float32))


def test_replay_add_converts_dtypes_safely(T):
    """DeviceReplay.add with int64 actions and bool dones (what argmax / comparisons give): the converted
    temporaries must stay alive until the push is queued — the ring holds the actions, not the dones."""
    tv, _ = T
    rb = tv.DeviceReplay(4096, 144, seed=3)
    n = 3000
    s = _synthetic_rows(0, n, 144, 0)
    a = torch.arange(n, device="cuda") % 4                                   # int64
    d = (torch.arange(n, device="cuda") % 5 == 0)                            # bool
    r = torch.arange(n, device="cuda", dtype=torch.float64)                  # float64
    rb.add(s, a, r, s, d)
    st, sa, sr, s2, sd = rb.sample(n, channels=3, side=12)
    idx = rb.last_indices(n)
    assert torch.equal(idx.sort().values, torch.arange(n, device="cuda"))   # a permutation of the filled slots
    assert torch.equal(sa.ravel(), idx % 4) and torch.equal(sd.ravel(), (idx % 5 == 0).float())
    assert torch.equal(sr.ravel(), idx.float())


@pytest.mark.parametrize("size,batch", [(65, 64), (68, 64), (1000, 999), (5000, 4096), (70000, 65536), (1 << 20, 4096)])
def test_sampler_is_without_replacement_at_any_batch(T, size, batch):
    """random.sample semantics (DDQN.py:191-200): `batch` distinct slots in [0, size), also right above
    size == batch (the first learn steps of Agent.step) and far above the old 1 024 limit."""
    tv, _ = T
    rb = tv.DeviceReplay(size, 16, seed=size)
    z = torch.zeros(size, 16, dtype=torch.int8, device="cuda")
    zz = torch.zeros(size, device="cuda")
    rb.add(z, zz.to(torch.int8), zz, z, zz.to(torch.int8))
    seen = None
    for _ in range(3):
        rb.sample(batch, channels=3, side=4)
        idx = rb.last_indices(batch)
        assert int(idx.min()) >= 0 and int(idx.max()) < size
        assert idx.unique().numel() == batch
        if seen is not None and batch < size - 8:
            assert not torch.equal(seen, idx)             # a fresh permutation per call
        seen = idx.clone()
    from tron import _native as nat
    with pytest.raises(nat.TronNativeError):               # random.sample raises when k > len(memory)
        rb.sample(size + 1, channels=3, side=4)


def test_sampler_uniformity(T):
    """Chi-square of slot counts and of (first draw, second draw) pair cells over many calls."""
    tv, _ = T
    size, batch, calls = 1000, 64, 400
    rb = tv.DeviceReplay(size, 16, seed=5)
    z = torch.zeros(size, 16, dtype=torch.int8, device="cuda")
    zz = torch.zeros(size, device="cuda")
    rb.add(z, zz.to(torch.int8), zz, z, zz.to(torch.int8))
    counts = np.zeros(size)
    pos_counts = np.zeros((4, 10))                         # position j in {0,1,31,63} x decile of the slot
    for _ in range(calls):
        rb.sample(batch, channels=3, side=4)
        idx = np_(rb.last_indices(batch))
        counts[idx] += 1
        for k, j in enumerate((0, 1, 31, 63)):
            pos_counts[k, idx[j] * 10 // size] += 1
    e = calls * batch / size
    chi2 = ((counts - e) ** 2 / e).sum()                   # 999 dof (hypergeometric draws: slightly under)
    assert 800 < chi2 < 1200, chi2
    chi2p = ((pos_counts - calls / 10) ** 2 / (calls / 10)).sum(1)   # 9 dof each
    assert np.all(chi2p < 30), chi2p


def test_acktr_rollout_at_config5_against_the_oracle(T):
    """ACKTR.train's rollout side (ACKTR.py:285-353: act, step, reward, mask = 1 - done, the stored observation replaced by the
    fresh game's when a game ended, the per-step env vector get_multy) at BASELINE config 5 — 16 384 envs of 32x32, temper mode,
    5 steps, both players — replayed on the CPU oracle with the actions the nets sampled: every stored observation plane, reward,
    mask, action and env vector, bit for bit, and the n-step returns (ACKTR.py:60-69) recomputed in numpy."""
    import ACKTR
    import config
    tv, oracle = T
    N, W, K = 16384, 32, 5
    rec = {"actions": [], "roll": None}

    def trace(kind, it, *a):
        if kind == "step":
            rec["actions"].append(np_(a[1]).copy())
        elif rec["roll"] is None:
            rolls, nxt = a
            rec["roll"] = [dict(obs=np_(r.observations), masks=np_(r.masks), rewards=np_(r.rewards), actions=np_(r.actions),
                                probs=np_(r.probs), returns=np_(r.returns)) for r in rolls]
            rec["next"] = [np_(v) for v in nxt]
    out = ACKTR.train(n_envs=N, width=W, model="mul", reward="3", iterations=1, acktr=False, num_steps=K, gamemode="temper",
                      seed=0x5EED, trace=trace)
    assert out["env_steps"] == N * K and len(rec["actions"]) == K
    c = config.reward_cons3
    ref = oracle.VecOracle(N, W, mode="temper", seed=0x5EED, stream=0,
                           reward=dict(step=-1.0, win=float(c[0]), lose=float(c[1]), draw=0.0, step_is_index=0))
    ref.reset_all()
    S = W + 2
    planes = lambda codes: np.moveaxis(oracle.pop_up(codes.reshape(N, S, S)), 0, 1)          # [N, 3, S, S]
    for p in range(2):                             # observation 0: the reset boards as each player sees them (game.py:124-132)
        assert np.array_equal(rec["roll"][p]["obs"][0], planes(oracle.state_for_player(ref.grid, p + 1))), p
    ended = 0
    for k in range(K):
        probs = [np.stack([ref.degree.astype(np.float32), ref.weight[:, p].astype(np.float32)], 1) for p in range(2)]
        a = rec["actions"][k]
        obs, done, winner, reward = ref.step(actions=a, autoreset=True)
        ended += int(done.sum())
        for p in range(2):
            r = rec["roll"][p]
            assert np.array_equal(r["obs"][k + 1], planes(obs[:, p])), (k, p)
            assert np.array_equal(r["rewards"][k][:, 0], reward[:, p]), (k, p)
            assert np.array_equal(r["masks"][k + 1][:, 0], 1.0 - done.astype(np.float32)), (k, p)
            assert np.array_equal(r["actions"][k][:, 0], a[:, p].astype(np.int64)), (k, p)
            assert np.array_equal(r["probs"][k], probs[p]), (k, p)
    assert ended == out["games"] and ended > 0
    for p in range(2):
        r = rec["roll"][p]
        ret = np.zeros_like(r["returns"])
        ret[-1] = rec["next"][p]
        for t in reversed(range(K)):
            ret[t] = ret[t + 1] * np.float32(ACKTR.GAMMA) * r["masks"][t + 1] + r["rewards"][t]
        assert np.allclose(r["returns"], ret, rtol=1e-6, atol=1e-6)
