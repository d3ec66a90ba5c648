"""Pins oracle/ (the C restatement) to the golden vectors that were produced by
running the reference (tests/golden/make_golden.py).  CPU only."""
import numpy as np
import pytest

import oracle
from conftest import load_golden


def test_philox_known_answers():
    # Random123 kat_vectors for philox4x32-10
    assert [int(x) for x in oracle.philox([0] * 4, [0] * 2)] == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    assert [int(x) for x in oracle.philox([0xffffffff] * 4, [0xffffffff] * 2)] == \
        [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    assert [int(x) for x in oracle.philox([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344],
                                          [0xa4093822, 0x299f31d0])] == \
        [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]


def test_step_exhaustive_4x4():
    g = load_golden("step_exhaustive_4x4")
    W = int(g["W"])
    n = len(g["starts"])
    for k in range(n):
        sg = oracle.ScalarGame(W, g["starts"][k])
        assert sg.step(int(g["actions"][k, 0]), int(g["actions"][k, 1])) == 0
        assert np.array_equal(sg.grid, g["grid"][k]), k
        assert np.array_equal(sg.pos, g["pos"][k]), k
        assert np.array_equal(sg.alive, g["alive"][k]), k
        assert int(sg.done[0]) == int(g["done"][k]) and int(sg.winner[0]) == int(g["winner"][k]), k
        assert np.array_equal(oracle.state_for_player(sg.grid, 1), g["obs1"][k]), k
        assert np.array_equal(oracle.state_for_player(sg.grid, 2), g["obs2"][k]), k


EPISODE_SETS = ["episodes_none_4", "episodes_none_10", "episodes_none_24", "episodes_none_32",
                "episodes_uniform_10", "episodes_ice_4", "episodes_ice_10", "episodes_ice_24",
                "episodes_temper_4", "episodes_temper_10", "episodes_temper_24"]


@pytest.mark.parametrize("name", EPISODE_SETS)
def test_episodes(name):
    g = load_golden(name)
    W = int(g["W"])
    mode = str(g["mode"])
    off = g["ep_off"]
    keep = "step_grid" in g.files
    for e in range(len(off) - 1):
        sg = oracle.ScalarGame(W, g["starts"][e], mode, g["slide"][e], g["weight"][e], g["degree"][e])
        for t in range(off[e], off[e + 1]):
            assert sg.step(int(g["actions"][t, 0]), int(g["actions"][t, 1]), g["uniforms"][t]) == 0
            assert np.array_equal(sg.pos, g["pos"][t]), (e, t)
            assert np.array_equal(sg.alive, g["alive"][t]), (e, t)
            assert int(sg.done[0]) == int(g["done"][t]), (e, t)
            assert np.array_equal(sg.consumed, g["consumed"][t]), (e, t)
            assert np.array_equal(sg.dir, g["actions"][t] + 1), (e, t)
            if keep:
                assert np.array_equal(sg.grid, g["step_grid"][t]), (e, t)
                assert np.array_equal(oracle.state_for_player(sg.grid, 1), g["step_obs1"][t])
                assert np.array_equal(oracle.state_for_player(sg.grid, 2), g["step_obs2"][t])
        assert int(sg.done[0]) == 1
        assert int(sg.winner[0]) == int(g["winner"][e]), e
        assert np.array_equal(sg.grid, g["final_grid"][e]), e
        assert np.array_equal(oracle.state_for_player(sg.grid, 1), g["final_obs1"][e])
        assert np.array_equal(oracle.state_for_player(sg.grid, 2), g["final_obs2"][e])
        assert sg.step(0, 0) == -1  # stepping a finished game is refused


def test_encode_and_popup():
    g = load_golden("encode")
    for W in (4, 10, 24):
        raw, codes, planes = g[f"raw_{W}"], g[f"codes_{W}"], g[f"planes_{W}"]
        for k in range(len(raw)):
            for p in (1, 2):
                c = oracle.state_for_player(raw[k], p)
                assert np.array_equal(c, codes[k, p - 1])
                assert np.array_equal(oracle.pop_up(c).astype(np.float64), planes[k, p - 1])


def test_env_scalars():
    g = load_golden("encode")
    for i, dg in enumerate(g["rate_degrees"]):
        assert oracle.get_rate(int(dg)) == g["rate_none"][i]
        for j, wt in enumerate(g["rate_weights"]):
            assert oracle.get_rate(int(dg), int(wt)) == g["rate"][i, j]   # bit-exact float64
    for s, d in zip(g["slides"], g["degree_slide"]):
        assert oracle.degree_slide(float(s)) == d
    assert np.all(g["prob_map_015"] == oracle.degree_slide(0.15))
    assert np.all(g["degree_map_m7"] == -7.0)
    assert list(g["multy0"]) == [-7.0, 55.0] and list(g["multy1"]) == [-7.0, 99.0]


@pytest.mark.parametrize("W", [4, 10, 24])
@pytest.mark.parametrize("mode", ["default", "fair"])
def test_make_game(W, mode):
    g = load_golden("reset")
    streams, res, grids = g[f"stream_{W}_{mode}"], g[f"result_{W}_{mode}"], g[f"grid_{W}_{mode}"]
    redraws = 0
    for k in range(len(streams)):
        start, weight, degree, n = oracle.make_game(W, mode == "fair", streams[k])
        assert list(start) == list(res[k, :4]), k
        assert list(weight) == list(res[k, 4:6]) and degree == res[k, 6], k
        assert n == res[k, 7], k
        assert np.array_equal(oracle.game_init(W, start), grids[k]), k
        redraws += n > (9 if mode == "fair" else 7)
    if W == 4 and mode == "default":
        assert redraws >= 60  # the forced-clash cases exercised the P1-only redraw rule


def test_rewards():
    g = load_golden("reward")["get_reward"]
    for ci, winner, win, lose, r1, r2 in g:
        table = dict(step=-1.0, win=win, lose=lose, draw=0.0, step_is_index=0)
        out = oracle.rewards(table, 1, int(winner))
        assert (float(out[0]), float(out[1])) == (r1, r2)
    # DDQN.py:289-305 and DQN.py:224-241 literal tables
    assert list(oracle.rewards(oracle.REWARD_DDQN, 0, 0)) == [-1, -1]
    assert list(oracle.rewards(oracle.REWARD_DDQN, 1, 1)) == [100, -100]
    assert list(oracle.rewards(oracle.REWARD_DDQN, 1, 2)) == [-100, 100]
    assert list(oracle.rewards(oracle.REWARD_DDQN, 1, 0)) == [0, 0]
    assert list(oracle.rewards(oracle.REWARD_DQN, 0, 0, 7)) == [7, 7]
    assert list(oracle.rewards(oracle.REWARD_DQN, 1, 2)) == [-25, 100]


def test_vec_matches_scalar_and_autoreset():
    """The batched driver (what the HIP VecTron is compared with) is the scalar
    step + make_game composed the way ACKTR.py:285-317 composes them."""
    N, W = 64, 6
    v = oracle.VecOracle(N, W, mode="ice", seed=123, stream=1, slide=0.3)
    v.reset_all()
    assert np.all(v.episode == 1)
    for t in range(40):
        grid0, pos0, w0, d0 = v.grid.copy(), v.pos.copy(), v.weight.copy(), v.degree.copy()
        tick0 = v.tick.copy()
        obs, done, winner, reward = v.step(autoreset=True)
        for i in range(N):
            x = oracle.philox([i, int(tick0[i]), 0, 0], [123, 1])
            sg = oracle.ScalarGame(W, pos0[i], "ice", 0.3, w0[i], d0[i])
            sg.grid[:] = grid0[i].reshape(W + 2, W + 2)
            u = [(int(x[2]) >> 8) / 16777216.0, (int(x[3]) >> 8) / 16777216.0]
            sg.step(int(x[0]) & 3, int(x[1]) & 3, u)
            assert int(done[i]) == int(sg.done[0]) and int(winner[i]) == int(sg.winner[0])
            if not done[i]:
                assert np.array_equal(v.grid[i], sg.grid.reshape(-1))
                assert np.array_equal(obs[i, 0], oracle.state_for_player(sg.grid, 1).reshape(-1))
            else:
                assert v.eplen[i] == 0 and v.done[i] == 0      # fresh game
                assert np.count_nonzero(v.grid[i] == 2) == 1 and np.count_nonzero(v.grid[i] == 4) == 1
                assert np.array_equal(obs[i, 1], oracle.state_for_player(v.grid[i], 2))
    assert v.episode.max() > 1


def test_minimax_matches_reference():
    """oracle.minimax_move == the reference's MinimaxPlayer.action on every recorded board:
    returned move, the root children's values, which children exist, and the draws consumed."""
    from golden.netgen import mm_stream
    g = load_golden("minimax")
    total = 0
    for W in g["widths"]:
        k = f"W{int(W)}_"
        for i in range(len(g[k + "move"])):
            stream = mm_stream(int(g[k + "seed"][i]), 96)
            move, values, expanded, used = oracle.minimax_move(g[k + "codes"][i], int(g[k + "depth"][i]),
                                                               int(g[k + "mode"][i]), stream)
            where = (int(W), i)
            assert np.array_equal(expanded, g[k + "expanded"][i].astype(bool)), where
            assert np.array_equal(values, g[k + "values"][i]), where
            assert move == int(g[k + "move"][i]), where
            assert used == int(g[k + "draws"][i]), where
            total += 1
    assert total > 2500


def test_vec_oracle_is_thread_count_invariant():
    """The OpenMP split of orc_vec_step (bench.py's all-cores cpu_baseline) changes nothing."""
    a = oracle.VecOracle(1500, 10, mode="temper", seed=7)
    b = oracle.VecOracle(1500, 10, mode="temper", seed=7)
    a.reset_all()
    b.reset_all()
    try:
        for _ in range(12):
            oracle.set_threads(1)
            ra = a.step(autoreset=True)
            oracle.set_threads(4)
            rb = b.step(autoreset=True)
            for x, y in zip(ra, rb):
                assert np.array_equal(x, y)
    finally:
        oracle.set_threads(1)
