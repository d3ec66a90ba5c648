"""GPU parity of the Minimax/Voronoi opponent (tron/minimax.py): the bitboard kernel against
(1) the searches recorded from the reference (tests/golden/minimax.npz) and (2) the literal
queue-and-tree oracle on random boards of every supported size.  Integer work: bit-exact."""
import ctypes as C

import numpy as np
import pytest

from conftest import load_golden
from golden.netgen import mm_stream

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

MODES = ("voronoi", "distwall")


@pytest.fixture(scope="module")
def T():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import tron.vec as tv
    import oracle
    return tv, oracle


def np_(t):
    return t.detach().cpu().numpy()


def bits(expanded):
    return np.stack([(expanded.astype(np.int32) >> a) & 1 for a in range(4)], 1).astype(bool)


def test_minimax_golden(T):
    """Every depth-2 search recorded from the reference: same root values, same searched moves, and
    the same move when the root's random.choice / randint draw is handed over."""
    tv, _ = T
    g = load_golden("minimax")
    checked = 0
    for W in g["widths"]:
        k = f"W{int(W)}_"
        for mode_id, mode in enumerate(MODES):
            sel = np.nonzero((g[k + "depth"] == 2) & (g[k + "mode"] == mode_id))[0]
            if len(sel) == 0:
                continue
            # the root draws last (minimax.py:270), so its draw is the last one the search consumed
            draws = np.array([mm_stream(int(g[k + "seed"][i]), 96)[int(g[k + "draws"][i]) - 1] for i in sel], np.int64)
            act, values, expanded = tv.minimax_codes(torch.from_numpy(g[k + "codes"][sel]).cuda(),
                                                     torch.from_numpy(draws).cuda(), mode)
            assert np.array_equal(bits(np_(expanded)), g[k + "expanded"][sel].astype(bool)), (int(W), mode)
            assert np.array_equal(np_(values), g[k + "values"][sel]), (int(W), mode)
            assert np.array_equal(np_(act).astype(np.int64) + 1, g[k + "move"][sel].astype(np.int64)), (int(W), mode)
            checked += len(sel)
    assert checked > 2000


def soup(rng, W, fill, close):
    """A random observation-code image with one +10 and one -10 head."""
    S = W + 2
    img = np.full((S, S), -1, np.int8)
    inner = np.where(rng.random((W, W)) < fill, rng.choice(np.array([-2, -3], np.int8), (W, W)), np.int8(1))
    img[1:-1, 1:-1] = inner
    r0, c0 = rng.integers(1, W + 1, 2)
    while True:
        if close:
            r1, c1 = r0 + rng.integers(-2, 3), c0 + rng.integers(-2, 3)
        else:
            r1, c1 = rng.integers(1, W + 1, 2)
        if (r1, c1) != (r0, c0) and 1 <= r1 <= W and 1 <= c1 <= W:
            break
    img[r0, c0], img[r1, c1] = 10, -10
    return img


def maze(W, flip):
    """Serpentine corridors: the longest flood fills a board of this size can have."""
    S = W + 2
    img = np.full((S, S), -1, np.int8)
    img[1:-1, 1:-1] = 1
    for r in range(2, W + 1, 2):
        img[r, 1:-1] = -2
        img[r, W if (r // 2) % 2 else 1] = 1
    img[1, 1], img[1, 3 if W >= 3 else 2] = (10, -10) if not flip else (-10, 10)
    return img


@pytest.mark.parametrize("W", [3, 8, 10, 14, 15, 24, 30, 31, 32, 33, 47, 62])      # 16 / 32 / 64 lanes per board: S <= 16, 32, 64
def test_minimax_random_boards_vs_oracle(T, W):
    tv, oracle = T
    rng = np.random.default_rng(1000 + W)
    n = 64 if W <= 33 else 24
    imgs = [soup(rng, W, rng.choice([0.0, 0.1, 0.3, 0.5, 0.7]), i % 3 == 0) for i in range(n)]
    imgs += [maze(W, False), maze(W, True)]
    imgs = np.stack(imgs)
    draws = rng.integers(0, 2 ** 32, len(imgs), dtype=np.int64)
    for mode_id, mode in enumerate(MODES):
        act, values, expanded = tv.minimax_codes(torch.from_numpy(imgs).cuda(), torch.from_numpy(draws).cuda(), mode)
        act, values, expanded = np_(act), np_(values), bits(np_(expanded))
        for i in range(len(imgs)):
            move, ov, oe, _ = oracle.minimax_move(imgs[i], 2, mode_id, np.full(8, draws[i], np.uint32))
            assert np.array_equal(expanded[i], oe), (W, mode, i)
            assert np.array_equal(values[i], ov), (W, mode, i)
            assert int(act[i]) + 1 == move, (W, mode, i)


def test_minimax_fuzz_mixed_batches(T):
    """Property test: batches whose boards differ wildly in fill and head distance share wavefronts (2 or
    4 boards per wave) — each board's result must not depend on its neighbours in the batch."""
    pytest.importorskip("hypothesis")
    from hypothesis import given, settings, strategies as st
    tv, oracle = T

    @settings(max_examples=20, deadline=None, derandomize=True)
    @given(W=st.integers(3, 34), n=st.integers(1, 9), seed=st.integers(0, 2 ** 31 - 1), mode_id=st.integers(0, 1))
    def run(W, n, seed, mode_id):
        rng = np.random.default_rng(seed)
        imgs = np.stack([soup(rng, W, rng.choice([0.0, 0.2, 0.6, 0.9]), bool(rng.integers(2))) for _ in range(n)])
        if n > 2:
            imgs[1] = -1                                       # a board without heads in the middle of a wave
        draws = rng.integers(0, 2 ** 32, n, dtype=np.int64)
        act, values, expanded = tv.minimax_codes(torch.from_numpy(imgs).cuda(), torch.from_numpy(draws).cuda(), MODES[mode_id])
        act, values, expanded = np_(act), np_(values), bits(np_(expanded))
        for i in range(n):
            if n > 2 and i == 1:
                assert act[i] == -1 and not expanded[i].any()
                continue
            move, ov, oe, _ = oracle.minimax_move(imgs[i], 2, mode_id, np.full(8, draws[i], np.uint32))
            assert np.array_equal(expanded[i], oe) and np.array_equal(values[i], ov) and int(act[i]) + 1 == move, (W, n, i)

    run()


@pytest.mark.parametrize("kw", [dict(mode=None), dict(mode=None, obs_is_state=False), dict(mode="ice", slide=0.3)],
                         ids=["obs-state", "grid", "ice"])
def test_minimax_env_actions(T, kw):
    """tron_minimax_actions on live envs == the oracle on each env's observation, with the Philox
    draw (env, tick, purpose 4, player)."""
    tv, oracle = T
    N, W, seed, rank = 384, 10, 0xBEEF, 3
    env = tv.VecTron(N, W, seed=seed, rank=rank, obs_format="codes", **kw)
    env.reset()
    for _ in range(7):
        env.step(None)
    for player in (1, 2):
        obs = np_(env.encode())[:, player - 1]
        tick = np_(env.state()["counters"])[:, 0]
        act, values, expanded = env.minimax_actions(player, want_values=True)
        act, values, expanded = np_(act), np_(values), bits(np_(expanded))
        for i in range(N):
            u = oracle.philox([i, int(tick[i]), 4, player], [seed, rank])[0]
            move, ov, oe, _ = oracle.minimax_move(obs[i], 2, 0, np.full(8, u, np.uint32))
            assert np.array_equal(expanded[i], oe) and np.array_equal(values[i], ov), (player, i)
            assert int(act[i]) + 1 == move, (player, i)


def test_minimax_beats_random(T):
    """Sanity at scale: player 2 driven by the search wins most games against uniform moves."""
    tv, _ = T
    env = tv.VecTron(4096, 10, seed=11, obs_format="codes", reward="ddqn")
    env.reset()
    gen = torch.Generator(device="cuda").manual_seed(5)
    wins = torch.zeros(3, dtype=torch.int64, device="cuda")
    for _ in range(60):
        a = torch.randint(0, 4, (env.N, 2), dtype=torch.int8, device="cuda", generator=gen)
        a[:, 1] = env.minimax_actions(2)
        _, _, done, winner = env.step(a)
        wins += torch.bincount(winner[done.bool()].long(), minlength=3)
    w = wins.cpu().numpy()
    assert w.sum() > 4096 and w[2] > 4 * w[1], w


def test_minimax_abi_errors(T):
    tv, _ = T
    import tron._native as nat
    L = nat.lib()
    codes = torch.full((1, 12, 12), -1, dtype=torch.int8, device="cuda")
    act = torch.zeros(1, dtype=torch.int8, device="cuda")
    args = lambda depth, mode, side: (nat.ptr(codes), 1, side, depth, mode, None, nat.ptr(act), None, None, nat.stream_ptr())
    assert L.tron_minimax_codes(*args(3, 0, 12)) == nat.ERR_UNSUPPORTED
    assert L.tron_minimax_codes(*args(2, 2, 12)) == nat.ERR_BAD_ARG
    assert L.tron_minimax_codes(*args(2, 0, 65)) == nat.ERR_BAD_ARG
    assert L.tron_minimax_codes(None, 1, 12, 2, 0, None, nat.ptr(act), None, None, nat.stream_ptr()) == nat.ERR_BAD_ARG
    # an image without heads: the move is -1, nothing is searched
    a, v, e = tv.minimax_codes(codes)
    torch.cuda.synchronize()
    assert int(a[0]) == -1 and int(e[0]) == 0
    big = tv.VecTron(2, 96)
    assert L.tron_minimax_actions(big._h, 1, 2, 0, nat.ptr(act), None, None, nat.stream_ptr()) == nat.ERR_UNSUPPORTED
    env = tv.VecTron(2, 10)
    assert L.tron_minimax_actions(env._h, 3, 2, 0, nat.ptr(act), None, None, nat.stream_ptr()) == nat.ERR_BAD_ARG
