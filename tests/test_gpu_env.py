"""GPU parity: the HIP path (through the C ABI) against (1) the golden vectors recorded
from the reference and (2) the CPU oracle on the same seeds.  Bit-exact everywhere —
this is integer/byte work; the only floats (rewards, pop_up planes) are exact small
integers.  Run with `pytest -m gpu` on the MI355X box."""
import numpy as np
import pytest

from conftest import load_golden

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")


@pytest.fixture(scope="module")
def T():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import tron.vec as tv
    import oracle
    return tv, oracle


def np_(t):
    return t.detach().cpu().numpy()


# --------------------------------------------------------------------- golden --
def test_golden_step_exhaustive_4x4(T):
    tv, _ = T
    g = load_golden("step_exhaustive_4x4")
    n = len(g["starts"])
    env = tv.VecTron(n, 4, obs_format="codes")
    env.reset(start_pos=torch.from_numpy(g["starts"]))
    obs, reward, done, winner = env.step(torch.from_numpy(g["actions"]), autoreset=False)
    st = env.state()
    assert np.array_equal(np_(env.grid()), g["grid"])
    assert np.array_equal(np_(st["pos"]), g["pos"])
    assert np.array_equal(np_(st["alive"]), g["alive"])
    assert np.array_equal(np_(done), g["done"]) and np.array_equal(np_(winner), g["winner"])
    assert np.array_equal(np_(st["dir"]), g["actions"] + 1)
    assert np.array_equal(np_(obs[:, 0]), g["obs1"]) and np.array_equal(np_(obs[:, 1]), g["obs2"])
    # DDQN reward table (DDQN.py:289-305)
    r = np_(reward)
    exp = np.where(g["done"][:, None] == 0, -1.0,
                   np.stack([np.select([g["winner"] == 1, g["winner"] == 2], [100.0, -100.0], 0.0),
                             np.select([g["winner"] == 2, g["winner"] == 1], [100.0, -100.0], 0.0)], 1))
    assert np.array_equal(r, exp.astype(np.float32))


EPISODE_SETS = ["episodes_none_4", "episodes_none_10", "episodes_none_24", "episodes_none_32",
                "episodes_uniform_10", "episodes_ice_4", "episodes_ice_10", "episodes_ice_24",
                "episodes_temper_4", "episodes_temper_10", "episodes_temper_24"]


@pytest.mark.parametrize("name", EPISODE_SETS)
def test_golden_episodes(T, name):
    """Every recorded reference episode is one env of a batch; finished envs idle (no autoreset)."""
    tv, _ = T
    g = load_golden(name)
    W, mode = int(g["W"]), str(g["mode"])
    off = g["ep_off"].astype(np.int64)
    n = len(off) - 1
    lens = np.diff(off)
    keep = "step_grid" in g.files
    env = tv.VecTron(n, W, mode=None if mode == "none" else mode, obs_format="codes")
    env.set_slide(torch.from_numpy(g["slide"]))
    env.reset(start_pos=torch.from_numpy(g["starts"]), weight=torch.from_numpy(g["weight"]),
              degree=torch.from_numpy(g["degree"]))
    for t in range(int(lens.max())):
        active = lens > t
        rows = np.where(active, off[:-1] + t, 0)
        a = np.where(active[:, None], g["actions"][rows], 0).astype(np.int8)
        u = np.where(active[:, None], g["uniforms"][rows], 0).astype(np.float32)
        obs, reward, done, winner = env.step(torch.from_numpy(a), torch.from_numpy(u), autoreset=False)
        st = env.state()
        act = np.nonzero(active)[0]
        assert np.array_equal(np_(st["pos"])[act], g["pos"][rows[act]]), t
        assert np.array_equal(np_(st["alive"])[act], g["alive"][rows[act]]), t
        assert np.array_equal(np_(done)[act], g["done"][rows[act]]), t
        if keep:
            assert np.array_equal(np_(env.grid())[act], g["step_grid"][rows[act]]), t
            assert np.array_equal(np_(obs[:, 0])[act], g["step_obs1"][rows[act]]), t
            assert np.array_equal(np_(obs[:, 1])[act], g["step_obs2"][rows[act]]), t
    assert np.all(np_(env.done) == 1)
    assert np.array_equal(np_(env.winner), g["winner"])
    assert np.array_equal(np_(env.grid()), g["final_grid"])
    obs = env.encode()
    assert np.array_equal(np_(obs[:, 0]), g["final_obs1"]) and np.array_equal(np_(obs[:, 1]), g["final_obs2"])


def test_golden_encode_stateless(T):
    tv, _ = T
    g = load_golden("encode")
    for W in (4, 10, 24):
        raw = torch.from_numpy(g[f"raw_{W}"]).cuda()
        for p in (1, 2):
            codes = tv.encode_codes(raw, p)
            assert np.array_equal(np_(codes), g[f"codes_{W}"][:, p - 1])
            planes = tv.pop_up_planes(codes)
            assert np.array_equal(np_(planes).astype(np.float64), g[f"planes_{W}"][:, p - 1])


# --------------------------------------------------------------- HIP vs oracle --
def _compare_state(env, ref, tag):
    st = env.state()
    N = ref.N
    assert np.array_equal(np_(env.grid()).reshape(N, -1), ref.grid), tag
    assert np.array_equal(np_(st["pos"]), ref.pos), tag
    assert np.array_equal(np_(st["alive"]), ref.alive), tag
    assert np.array_equal(np_(st["dir"]), ref.dir), tag
    assert np.array_equal(np_(st["done"]), ref.done) and np.array_equal(np_(st["winner"]), ref.winner), tag
    assert np.array_equal(np_(st["weight"]), ref.weight) and np.array_equal(np_(st["degree"]), ref.degree), tag
    c = np_(st["counters"]).astype(np.uint32)
    assert np.array_equal(c[:, 0], ref.tick) and np.array_equal(c[:, 1], ref.episode), tag
    assert np.array_equal(c[:, 2], ref.eplen), tag


CASES = [
    # N,   W,  mode,     fair,  autoreset, reward      (mode None + even W runs the observation-is-state kernel)
    (64, 10, None, False, True, "ddqn"),
    (64, 10, None, False, True, "grid"),          # same, forced onto the board-owning kernel (k_tile)
    (70, 24, None, True, False, "grid"),
    (200, 6, None, True, False, "acktr"),         # observation-is-state, no autoreset, masked resets
    (300, 10, None, False, True, "inc"),          # incremental in-place update (TRON_STEP_INCREMENTAL)
    (77, 24, None, True, True, "inc"),
    (150, 8, None, False, False, "inc"),
    (100, 10, "temper", False, True, "acktr"),     # tail tile (100 = 64 + 36)
    (37, 4, "ice", True, True, "dqn"),             # fair starts, step-index reward
    (130, 7, "ice", False, True, "ddqn"),          # odd W: generic (G % 4 != 0) path
    (33, 5, "temper", True, False, "ddqn"),        # odd W, no autoreset
    (70, 24, None, False, True, "ddqn"),
    (40, 32, "temper", False, True, "ddqn"),       # E = 32 tiles
    (20, 47, "ice", False, True, "ddqn"),          # E = 16 tiles, odd W
]


@pytest.mark.parametrize("N,W,mode,fair,autoreset,reward", CASES)
def test_hip_vs_oracle_philox(T, N, W, mode, fair, autoreset, reward):
    tv, oracle = T
    ois = reward != "grid"
    inc = reward == "inc"
    reward = "ddqn" if reward in ("grid", "inc") else reward
    table = {"ddqn": oracle.REWARD_DDQN, "dqn": oracle.REWARD_DQN, "acktr": oracle.REWARD_ACKTR}[reward]
    env = tv.VecTron(N, W, mode=mode, fair=fair, seed=1234, rank=3, obs_format="codes", reward=reward, slide=0.3,
                     obs_is_state=ois, incremental=inc)
    assert env.obs_is_state == (ois and W % 2 == 0) and env.incremental == inc
    ref = oracle.VecOracle(N, W, mode=mode, seed=1234, stream=3, fair=fair, reward=table, slide=0.3)
    obs0 = env.reset()
    ref.reset_all()
    _compare_state(env, ref, "reset")
    o_ref = np.stack([[oracle.state_for_player(ref.grid[i], p) for p in (1, 2)] for i in range(N)])
    assert np.array_equal(np_(obs0).reshape(N, 2, -1), o_ref)
    steps = 30 if W <= 10 else 12
    for t in range(steps):
        obs, r, d, w = env.step(autoreset=autoreset)
        o, dd, ww, rr = ref.step(autoreset=autoreset)
        assert np.array_equal(np_(obs).reshape(N, 2, -1), o), t
        assert np.array_equal(np_(d), dd) and np.array_equal(np_(w), ww), t
        assert np.array_equal(np_(r), rr), t
        _compare_state(env, ref, t)
        if not autoreset and t % 5 == 4:          # DDQN-style: reset the finished envs explicitly
            mask = ref.done.copy()
            env.reset(mask=torch.from_numpy(mask))
            ref.reset_masked(mask)
            _compare_state(env, ref, ("masked reset", t))
    assert ref.episode.max() > 1


@pytest.mark.parametrize("N,W,mode,variant", [(300, 10, None, "obs"), (300, 10, None, "grid"), (260, 24, None, "inc"),
                                              (200, 8, "temper", "grid"), (90, 7, "ice", "grid")])
def test_nonreversing_policy_vs_oracle(T, N, W, mode, variant):
    """TRON_STEP_NONREVERSING: in-kernel actions drawn from the three headings that do not reverse the
    last move (SURVEY.md §8(d) secondary policy) — same draws as the oracle, in every kernel; and
    no player ever reverses."""
    tv, oracle = T
    env = tv.VecTron(N, W, mode=mode, seed=77, rank=1, obs_format="codes", slide=0.25,
                     obs_is_state=variant != "grid", incremental=variant == "inc")
    ref = oracle.VecOracle(N, W, mode=mode, seed=77, stream=1, slide=0.25)
    env.reset()
    ref.reset_all()
    lens = []
    for t in range(40):
        before = ref.dir.copy()
        obs, r, d, w = env.step(autoreset=True, nonreversing=True)
        o, dd, ww, rr = ref.step(autoreset=True, nonreversing=True)
        assert np.array_equal(np_(obs).reshape(N, 2, -1), o), t
        assert np.array_equal(np_(d), dd) and np.array_equal(np_(w), ww) and np.array_equal(np_(r), rr), t
        _compare_state(env, ref, t)
        lens.append(dd.mean())
        if mode is None:                           # no slides: the direction taken is the action drawn
            st = np_(env.state()["dir"])
            moved = (before > 0) & (dd[:, None] == 0)
            assert not np.any(moved & (((st - before) % 4) == 2)), t
    if mode is None:
        assert np.mean(lens) < 0.34                # fewer games end per step than under uniform actions (0.36-0.43)
    # through the rollout entry point as well
    env.rollout_random(5, nonreversing=True)
    for _ in range(5):
        ref.step(autoreset=True, nonreversing=True, want_obs=False)
    _compare_state(env, ref, "rollout")


def test_fuzz_hip_vs_oracle(T):
    """Property test: for arbitrary (N, W, mode, fair, autoreset, policy, kernel choice, seed) a few
    Philox-driven steps leave HIP and oracle in identical states with identical outputs."""
    hyp = pytest.importorskip("hypothesis")
    from hypothesis import given, settings, strategies as st
    tv, oracle = T

    @settings(max_examples=30, deadline=None, derandomize=True)
    @given(N=st.integers(1, 150), W=st.integers(2, 40), mode=st.sampled_from([None, "ice", "temper"]),
           fair=st.booleans(), autoreset=st.booleans(), nonrev=st.booleans(), ois=st.booleans(),
           seed=st.integers(0, 2 ** 32 - 1), slide=st.sampled_from([0.0, 0.15, 0.5, 1.0]))
    def run(N, W, mode, fair, autoreset, nonrev, ois, seed, slide):
        env = tv.VecTron(N, W, mode=mode, fair=fair, seed=seed, rank=5, obs_format="codes", slide=slide, obs_is_state=ois)
        ref = oracle.VecOracle(N, W, mode=mode, seed=seed, stream=5, fair=fair, slide=slide)
        env.reset()
        ref.reset_all()
        for t in range(6):
            obs, r, d, w = env.step(autoreset=autoreset, nonreversing=nonrev)
            o, dd, ww, rr = ref.step(autoreset=autoreset, nonreversing=nonrev)
            assert np.array_equal(np_(obs).reshape(N, 2, -1), o)
            assert np.array_equal(np_(d), dd) and np.array_equal(np_(w), ww) and np.array_equal(np_(r), rr)
        _compare_state(env, ref, (N, W, mode, fair, autoreset, nonrev, ois, seed))
        env.close()

    run()


def test_fuzz_formats_actions_resets(T):
    """Property test over the other entry points: f32 plane formats, the incremental kernel, explicit
    actions / slide uniforms, masked resets between steps, encode() into another format."""
    pytest.importorskip("hypothesis")
    from hypothesis import given, settings, strategies as st
    tv, oracle = T

    @settings(max_examples=30, deadline=None, derandomize=True)
    @given(N=st.integers(1, 90), W=st.integers(2, 34), mode=st.sampled_from([None, "ice", "temper"]),
           fmt=st.sampled_from(["codes", "planes3", "planes4", "inc"]), explicit=st.booleans(),
           seed=st.integers(0, 2 ** 32 - 1))
    def run(N, W, mode, fmt, explicit, seed):
        rng = np.random.RandomState(seed % (2 ** 31))
        inc = fmt == "inc"
        env = tv.VecTron(N, W, mode=mode, seed=seed, obs_format="codes" if inc else fmt, incremental=inc, slide=0.3)
        ref = oracle.VecOracle(N, W, mode=mode, seed=seed, slide=0.3)
        env.reset()
        ref.reset_all()
        S = W + 2
        for t in range(6):
            a = rng.randint(0, 4, size=(N, 2)).astype(np.int8) if explicit else None
            u = (rng.randint(0, 1 << 24, size=(N, 2)) / float(1 << 24)).astype(np.float32) if explicit else None
            obs, r, d, w = env.step(None if a is None else torch.from_numpy(a), None if u is None else torch.from_numpy(u),
                                    autoreset=False)
            o, dd, ww, rr = ref.step(a, u, autoreset=False)
            if fmt in ("codes", "inc"):
                assert np.array_equal(np_(obs).reshape(N, 2, -1), o)
            else:
                planes = np.stack([oracle.pop_up(o[i, p]) for i in range(N) for p in range(2)]).reshape(N, 2, 3, S, S)
                assert np.array_equal(np_(obs)[:, :, :3], planes)
                if fmt == "planes4":
                    assert np.all(np_(obs)[:, :, 3] == np.float32(oracle.degree_slide(0.3)))
            assert np.array_equal(np_(d), dd) and np.array_equal(np_(w), ww) and np.array_equal(np_(r), rr)
            if t % 2 == 1:                          # DDQN-style: restart some of the finished games by hand
                mask = (ref.done.astype(bool) & (rng.rand(N) < 0.7)).astype(np.int8)
                env.reset(mask=torch.from_numpy(mask))
                ref.reset_masked(mask)
        _compare_state(env, ref, (N, W, mode, fmt, explicit, seed))
        codes = env.encode("codes")
        assert np.array_equal(np_(codes).reshape(N, 2, -1),
                              np.stack([[oracle.state_for_player(ref.grid[i], p) for p in (1, 2)] for i in range(N)]))
        env.close()

    run()


@pytest.mark.parametrize("W,mode", [(10, None), (6, "ice"), (9, "temper")])
def test_explicit_actions_and_planes(T, W, mode):
    """Caller-supplied actions/uniforms + the f32 plane formats (pop_up, prob_map plane)."""
    tv, oracle = T
    N = 96
    rng = np.random.RandomState(5)
    slide = rng.choice([0.0, 0.03, 0.15, 0.36], size=N)
    env3 = tv.VecTron(N, W, mode=mode, seed=9, obs_format="planes3")
    env4 = tv.VecTron(N, W, mode=mode, seed=9, obs_format="planes4")
    ref = oracle.VecOracle(N, W, mode=mode, seed=9)
    ref.slide[:] = slide
    for e in (env3, env4):
        e.set_slide(torch.from_numpy(slide))
        e.reset()
    ref.reset_all()
    for t in range(15):
        a = rng.randint(0, 4, size=(N, 2)).astype(np.int8)
        u = (rng.randint(0, 1 << 24, size=(N, 2)) / float(1 << 24)).astype(np.float32)
        o3, r3, d3, w3 = env3.step(torch.from_numpy(a), torch.from_numpy(u), autoreset=True)
        o4, r4, d4, w4 = env4.step(torch.from_numpy(a), torch.from_numpy(u), autoreset=True)
        o, dd, ww, rr = ref.step(a, u, autoreset=True)
        planes = np.stack([oracle.pop_up(o[i, p]) for i in range(N) for p in range(2)]).reshape(N, 2, 3, W + 2, W + 2)
        assert np.array_equal(np_(o3), planes), t
        assert np.array_equal(np_(o4)[:, :, :3], planes), t
        p4 = np.array([np.float32(oracle.degree_slide(s)) for s in slide], np.float32)
        assert np.array_equal(np_(o4)[:, :, 3], np.broadcast_to(p4[:, None, None, None], (N, 2, W + 2, W + 2))), t
        assert np.array_equal(np_(d3), dd) and np.array_equal(np_(w4), ww) and np.array_equal(np_(r3), rr), t
    # encode() in another format gives the same state
    codes = env3.encode("codes")
    assert np.array_equal(np_(codes).reshape(N, 2, -1),
                          np.stack([[oracle.state_for_player(ref.grid[i], p) for p in (1, 2)] for i in range(N)]))


def test_step_without_obs_and_totals(T):
    tv, oracle = T
    N, W = 256, 10
    env = tv.VecTron(N, W, seed=77, obs_format=None)
    ref = oracle.VecOracle(N, W, seed=77)
    env.reset()
    ref.reset_all()
    totals = torch.zeros(4, dtype=torch.int64, device="cuda")
    env.rollout_random(25, totals)
    exp = np.zeros(4, np.int64)
    for _ in range(25):
        _, d, w, _ = ref.step(autoreset=True, want_obs=False)
        exp += [N, int(((d == 1) & (w == 1)).sum()), int(((d == 1) & (w == 2)).sum()), int(((d == 1) & (w == 0)).sum())]
    assert np.array_equal(np_(totals), exp)
    _compare_state(env, ref, "rollout")


@pytest.mark.parametrize("resident", [False, True])
@pytest.mark.parametrize("mode,slide", [(None, 0.15), ("temper", 0.15), ("ice", 0.4)])
@pytest.mark.parametrize("N,W,K", [(300, 10, 37), (5000, 24, 70), (33, 2, 9), (70, 16, 130)])
def test_persistent_rollout_equals_stepwise(T, N, W, K, resident, mode, slide):
    """tron_rollout_random on the observation-is-state path is ONE launch in which every workgroup steps
    its own tiles K times (k_obs_roll; the sliding modes: k_obs_roll_slide, their slide tiles kept in the per-env log that
    grid() replays): same state — board image included —, observations and totals as K oracle steps."""
    tv, oracle = T
    env = tv.VecTron(N, W, mode=mode, slide=slide, seed=2024, rank=6, obs_format="codes")
    assert env.obs_is_state
    ref = oracle.VecOracle(N, W, mode=mode, slide=slide, seed=2024, stream=6)
    env.reset()
    ref.reset_all()
    totals = torch.zeros(4, dtype=torch.int64, device="cuda")
    env.rollout_random(K, totals, resident=resident)
    exp = np.zeros(4, np.int64)
    for _ in range(K):
        o, d, w, _ = ref.step(autoreset=True)
        exp += [N, int(((d == 1) & (w == 1)).sum()), int(((d == 1) & (w == 2)).sum()), int(((d == 1) & (w == 0)).sum())]
    assert np.array_equal(np_(totals), exp)
    assert np.array_equal(np_(env.obs).reshape(N, 2, -1), o)
    assert np.array_equal(np_(env.grid()).reshape(N, -1), ref.grid)       # slide tiles as slide tiles
    if mode is not None and K > 30 and W >= 10:
        assert int(np.isin(ref.grid, (5, 6)).sum()) > 0                  # (and there are some)
    _compare_state(env, ref, "persistent rollout")
    env.step()                                      # and the per-step kernel carries on from there
    ref.step(autoreset=True)
    _compare_state(env, ref, "step after rollout")


@pytest.mark.parametrize("N,W,mode,nparts", [(300, 10, None, 2), (5000, 24, None, 3), (700, 10, "temper", 2), (97, 7, "ice", 4)])
def test_step_in_slices_equals_whole_step(T, N, W, mode, nparts):
    """tron_step_encode_part: the env tiles stepped slice by slice — each slice on a stream of its own, as a caller
    pipelining the slices against its policy would — leave the same state and outputs as one whole step."""
    tv, oracle = T
    env = tv.VecTron(N, W, mode=mode, seed=31, rank=5, obs_format="codes")
    ref = oracle.VecOracle(N, W, mode=mode, seed=31, stream=5)
    env.reset()
    ref.reset_all()
    ranges = [env.part_range(p, nparts) for p in range(nparts)]
    assert ranges[0][0] == 0 and sum(n for _, n in ranges) == N
    assert all(ranges[i][0] + ranges[i][1] == ranges[i + 1][0] for i in range(nparts - 1))
    streams = [torch.cuda.Stream() for _ in range(nparts)]
    torch.cuda.synchronize()
    rs = np.random.RandomState(3)
    for t in range(6):
        acts = None if t % 2 else torch.from_numpy(rs.randint(0, 4, (N, 2)).astype(np.int8)).cuda()
        torch.cuda.synchronize()
        for p in reversed(range(nparts)):                       # any order, any stream
            with torch.cuda.stream(streams[p]):
                env.step_part(p, nparts, acts)
        torch.cuda.synchronize()
        o, d, w, r = ref.step(None if acts is None else np_(acts), autoreset=True)
        assert np.array_equal(np_(env.obs).reshape(N, 2, -1), o), t
        assert np.array_equal(np_(env.done), d) and np.array_equal(np_(env.winner), w) and np.array_equal(np_(env.reward), r)
    _compare_state(env, ref, "sliced steps")


@pytest.mark.parametrize("N,W,mode,K", [(65536, 24, None, 9), (4096, 10, None, 20), (2000, 10, "temper", 11)])
def test_two_stream_rollout_equals_oracle(T, N, W, mode, K):
    """TRON_ROLLOUT_TWO_STREAMS: one launch per step and per half of the envs, the halves on two streams."""
    tv, oracle = T
    import os
    oracle.set_threads(min(16, len(os.sched_getaffinity(0))))
    env = tv.VecTron(N, W, mode=mode, seed=8, rank=1, obs_format="codes")
    ref = oracle.VecOracle(N, W, mode=mode, seed=8, stream=1)
    env.reset()
    ref.reset_all()
    totals = torch.zeros(4, dtype=torch.int64, device="cuda")
    env.rollout_random(K, totals, two_streams=True)
    torch.cuda.synchronize()
    for k in range(K):
        o, _, _, _ = ref.step(autoreset=True, want_obs=(k == K - 1))
    oracle.set_threads(1)
    assert int(totals[0]) == N * K
    assert np.array_equal(np_(env.obs).reshape(N, 2, -1), o)
    _compare_state(env, ref, "two-stream rollout")


@pytest.mark.parametrize("N,W,K,kw", [(260, 10, 70, dict(mode="temper", obs_is_state=False)), (90, 7, 33, dict(mode="ice", slide=0.4)),
                                      (150, 9, 20, dict(obs_format="planes3")), (64, 12, 65, dict(obs_format="planes4", mode="temper")),
                                      (100, 10, 40, dict(obs_is_state=False))])
def test_persistent_rollout_board_layout(T, N, W, K, kw):
    """The same for the board-owning layout (k_tile_roll): every mode, format and side."""
    tv, oracle = T
    fmt = kw.get("obs_format", "codes")
    env = tv.VecTron(N, W, seed=99, rank=2, **dict(dict(obs_format="codes"), **kw))
    assert not env.obs_is_state
    ref = oracle.VecOracle(N, W, mode=kw.get("mode"), seed=99, stream=2, slide=kw.get("slide", 0.15))
    env.reset()
    ref.reset_all()
    env.rollout_random(K)
    for _ in range(K):
        o, _, _, _ = ref.step(autoreset=True)
    _compare_state(env, ref, "board rollout")
    if fmt == "codes":
        assert np.array_equal(np_(env.obs).reshape(N, 2, -1), o)
    else:
        planes = np.stack([oracle.pop_up(o[i, p]) for i in range(N) for p in range(2)]).reshape(N, 2, 3, W + 2, W + 2)
        assert np.array_equal(np_(env.obs)[:, :, :3], planes)


# ----------------------------------------------------- full size: properties --
@pytest.mark.parametrize("N,W,mode,steps", [
    (1, 2, None, 6),            # smallest board, a single env (BASELINE configs[0] shape is N=1)
    (70, 2, None, 6),           # 16-cell boards: ONE 16-byte chunk per env (found by the fuzz test: chunk -> env
    (70, 2, "ice", 6),          #   mapping by multiplication needs its own case when cpe == 1)
    (1, 10, "temper", 20),
    (3, 96, None, 6),           # largest supported side: 98x98 cells, LDS tile of 4 envs
    (5, 95, "ice", 5),          # largest odd side (bytewise global access)
    (65, 16, None, 10),         # one env past a tile boundary
    (4096, 10, None, 25),       # BASELINE configs[1] env shape
    (4096, 10, "temper", 25),
])
def test_extreme_and_config_shapes_vs_oracle(T, N, W, mode, steps):
    tv, oracle = T
    env = tv.VecTron(N, W, mode=mode, seed=42, rank=1, obs_format="codes", reward="acktr")
    ref = oracle.VecOracle(N, W, mode=mode, seed=42, stream=1, reward=oracle.REWARD_ACKTR)
    env.reset()
    ref.reset_all()
    for t in range(steps):
        obs, r, d, w = env.step()
        o, dd, ww, rr = ref.step(autoreset=True)
        assert np.array_equal(np_(obs).reshape(N, 2, -1), o), t
        assert np.array_equal(np_(d), dd) and np.array_equal(np_(w), ww) and np.array_equal(np_(r), rr), t
    _compare_state(env, ref, "end")


def test_config5_shape_16384x32_temper_prefix(T):
    """BASELINE configs[4] env shape (ACKTR: 16 384 envs, 32x32, temper, f32 planes): the first 1024
    envs equal the oracle on the same seed; planes equal pop_up of the codes everywhere."""
    tv, oracle = T
    N, W, NP = 16384, 32, 1024
    env = tv.VecTron(N, W, mode="temper", seed=7, obs_format="planes3", reward="acktr")
    ref = oracle.VecOracle(NP, W, mode="temper", seed=7, reward=oracle.REWARD_ACKTR)
    env.reset()
    ref.reset_all()
    for _ in range(8):
        obs, r, d, w = env.step()
        o, dd, ww, rr = ref.step(autoreset=True)
    assert np.array_equal(np_(d[:NP]), dd) and np.array_equal(np_(r[:NP]), rr)
    assert np.array_equal(np_(env.grid()[:NP]).reshape(NP, -1), ref.grid)
    codes = env.encode("codes")
    assert np.array_equal(np_(codes[:NP]).reshape(NP, 2, -1), o)
    assert torch.equal(tv.pop_up_planes(codes.reshape(2 * N, W + 2, W + 2)).view(N, 2, 3, W + 2, W + 2), obs)


def test_full_size_incremental_matches_full_rewrite(T):
    """65 536 x 24x24: the in-place incremental step and the full-rewrite step stay identical."""
    tv, _ = T
    N, W = 65536, 24
    a = tv.VecTron(N, W, seed=99, obs_format="codes")
    b = tv.VecTron(N, W, seed=99, obs_format="codes", incremental=True)
    a.reset()
    b.reset()
    for _ in range(12):
        oa, ra, da, wa = a.step()
        ob, rb, db, wb = b.step()
    assert torch.equal(oa, ob) and torch.equal(ra, rb) and torch.equal(da, db) and torch.equal(wa, wb)
    sa, sb = a.state(), b.state()
    assert all(torch.equal(sa[k], sb[k]) for k in sa)
    assert torch.equal(a.grid(), b.grid())


def test_full_size_65536x24_properties(T):
    """BASELINE config 3 size.  Size-independent checks: (a) the first 2048 envs equal the
    oracle run on the same seed (Philox is keyed by env index, so a prefix is self-contained);
    (b) obs is exactly the stateless encode of the grid; (c) per-env tile census matches
    the counters: one head per player, bodies == eplen, intact border for live envs."""
    tv, oracle = T
    N, W, K, NP = 65536, 24, 16, 2048
    env = tv.VecTron(N, W, seed=0x5EED, obs_format="codes")
    ref = oracle.VecOracle(NP, W, seed=0x5EED)
    env.reset()
    ref.reset_all()
    for _ in range(K):
        obs, r, d, w = env.step(autoreset=True)
        o, dd, ww, rr = ref.step(autoreset=True)
    assert np.array_equal(np_(obs[:NP]).reshape(NP, 2, -1), o)
    assert np.array_equal(np_(d[:NP]), dd) and np.array_equal(np_(w[:NP]), ww)
    grid = env.grid()
    assert np.array_equal(np_(grid[:NP]).reshape(NP, -1), ref.grid)
    for p in (1, 2):
        assert torch.equal(tv.encode_codes(grid, p), obs[:, p - 1])
    st = env.state()
    eplen = st["counters"][:, 2].to(torch.int64)
    flat = grid.reshape(N, -1)
    assert torch.all((flat == 2).sum(1) == 1) and torch.all((flat == 4).sum(1) == 1)
    assert torch.equal((flat == 1).sum(1), eplen) and torch.equal((flat == 3).sum(1), eplen)
    S = W + 2
    border = torch.ones(S, S, dtype=torch.bool, device="cuda")
    border[1:-1, 1:-1] = False
    assert torch.all(grid[:, border] == -1)          # autoreset leaves only live boards behind
    assert int(st["counters"][:, 1].max()) > 1


# ----------------------------------------------------------------- ABI errors --
def test_abi_error_codes(T):
    import ctypes as C
    from tron import _native as nat
    L = nat.lib()
    h = C.c_void_p()
    assert L.tron_create(0, 10, 0, 0, 1, 0, C.byref(h)) == -1          # n_envs < 1
    assert L.tron_create(8, 1, 0, 0, 1, 0, C.byref(h)) == -1           # W too small
    assert L.tron_create(8, 10, 7, 0, 1, 0, C.byref(h)) == -1          # bad mode
    assert L.tron_step_encode(None, None, None, 0, 0, None, None, None, None, None) == -1
    assert L.tron_create(8, 10, 0, 0, 1, 0, C.byref(h)) == 0
    assert L.tron_step_encode(h, None, None, 0, 1, None, None, None, None, None) == -1   # fmt without buffer
    assert L.tron_step_encode(h, None, None, 8, 0, None, None, None, None, None) == -1   # unknown flag
    assert L.tron_encode(h, 0, None, None) == -1
    assert L.tron_destroy(h) == 0
    assert nat.lib().tron_strerror(-1) == b"bad argument"
