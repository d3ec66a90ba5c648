"""GPU tests of the reference-compatible host surface (Game / Map / pop_up / make_game /
Agent / ReplayBuffer) and of the device replay memory — all of it runs through the C ABI."""
import numpy as np
import pytest

from conftest import load_golden

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")


def test_game_facade_replays_reference_episodes():
    """BASELINE config 1 shape: a single 10x10 game driven through Game.step, checked against
    the episodes recorded from the reference's Game.step (positions, alive, done, winner, grid,
    both observations, history directions)."""
    from tron.game import Game, PositionPlayer
    from tron.player import ACPlayer, Direction
    g = load_golden("episodes_none_10")
    off = g["ep_off"]
    for e in range(12):
        s = g["starts"][e]
        game = Game(10, 10, [PositionPlayer(1, ACPlayer(), [int(s[0]), int(s[1])]),
                             PositionPlayer(2, ACPlayer(), [int(s[2]), int(s[3])])])
        assert game.winner is None and not game.done and len(game.history) == 1
        assert game.map()[int(s[0]), int(s[1])].value == 2 and game.map()[int(s[2]), int(s[3])].value == 4
        for t in range(off[e], off[e + 1]):
            a = g["actions"][t]
            n1, n2, done = game.step(int(a[0]), int(a[1]))
            assert n1.dtype == np.int64 and n1.shape == (12, 12)
            assert np.array_equal(n1, g["step_obs1"][t]) and np.array_equal(n2, g["step_obs2"][t])
            assert np.array_equal(game.history[-1].map.array(), g["step_grid"][t])
            assert [*game.pps[0].position, *game.pps[1].position] == list(g["pos"][t])
            assert [int(game.pps[0].alive), int(game.pps[1].alive)] == list(g["alive"][t])
            assert done == bool(g["done"][t])
            assert game.history[-2].player_one_direction == Direction(int(a[0]) + 1)
        assert game.done and (0 if game.winner is None else game.winner) == int(g["winner"][e])
        with pytest.raises(RuntimeError):
            game.step(0, 0)


def test_map_and_pop_up_facade():
    from tron.map import Map, Tile
    from tron.util import pop_up, prob_map, get_reward
    e = load_golden("encode")
    for k in range(4):
        m = Map.from_codes(10, e["raw_10"][k])
        for p in (1, 2):
            codes = m.state_for_player(p)
            assert codes.dtype == np.int64 and np.array_equal(codes, e["codes_10"][k, p - 1])
            planes = pop_up(codes)
            assert planes.dtype == np.float64 and np.array_equal(planes, e["planes_10"][k, p - 1])
    m = Map(4, 4, Tile.EMPTY, Tile.WALL)
    assert m[0, 0] is Tile.EMPTY and m.array()[0, 0] == -1
    m[1, 2] = Tile.PLAYER_TWO_slide
    assert m[1, 2] is Tile.PLAYER_TWO_slide and m.clone()[1, 2] is Tile.PLAYER_TWO_slide
    assert m.color(Tile.PLAYER_ONE_HEAD, 1) == 10 and m.color(Tile.PLAYER_ONE_HEAD, 2) == -10
    assert m.color(Tile.EMPTY, 1) == 1 and m.color(Tile.PLAYER_TWO_slide, 1) == -3
    assert np.array_equal(prob_map(3.0), e["util_prob_map_3"])
    r = load_golden("reward")["get_reward"]

    class G:
        winner = None
    for ci, w, win, lose, r1, r2 in r:
        G.winner = None if w == 0 else int(w)
        assert get_reward(G, [win, lose]) == (r1, r2)


def test_game_env_scalars_and_setters():
    from tron.util import make_game
    e = load_golden("encode")
    game = make_game(True, True, gamemode="temper")
    assert 40 <= game.weight[0] <= 101 and 40 <= game.weight[1] <= 101 and -30 <= game.degree <= 30
    assert game.pps[0].position != game.pps[1].position
    assert all(0 <= v < 10 for pp in game.pps for v in pp.position)
    game.weight = [55, 99]
    game.degree = -7
    game.slide = 0.15
    st = game._env.state()
    assert st["weight"][0].tolist() == [55, 99] and int(st["degree"][0]) == -7
    assert game.get_multy(0) == list(e["multy0"]) and game.get_multy(1) == list(e["multy1"])
    assert np.array_equal(game.prob_map(), e["prob_map_015"]) and np.array_equal(game.degree_map(), e["degree_map_m7"])
    i, j = list(e["rate_degrees"]).index(-7), list(e["rate_weights"]).index(55)
    assert game.get_rate(0) == e["rate"][i, j] and game.get_rate() == e["rate_none"][i]
    fair = make_game(True, True, mode="fair", gamemode="ice", slide_pram=0.3)
    assert fair.slide == 0.3 and fair.mode == "ice"
    from tron.minimax import MinimaxPlayer
    assert isinstance(make_game(False, True).pps[0].player, MinimaxPlayer)      # util.py:82


def test_main_loop_with_stub_model_and_ai_action():
    """Game.main_loop(model, pop) with a random-action stub (the pygame-free CPU loop of
    BASELINE config 1) terminates with a consistent winner; Ai.action returns a Direction."""
    import random
    from tron.util import make_game, pop_up
    from tron.player import Direction
    import DQN

    class Stub:
        def act(self, x, env=None):
            assert x.shape == (1, 3, 12, 12)
            return torch.tensor([random.randrange(4)])

    for _ in range(5):
        game = make_game(True, True)
        game.main_loop(Stub(), pop_up)
        alive = [pp.alive for pp in game.pps]
        assert game.done and sum(alive) <= 1
        if game.winner is not None:
            assert alive[game.winner - 1] and game.pps[0].position != game.pps[1].position
    game = make_game(True, True)
    ai = DQN.Ai(epsilon=0.5)
    assert isinstance(ai.action(game.map(), 1), Direction)


def test_main_loop_replays_reference_fixture():
    """tests/golden/main_loop.npz: the reference's Game.main_loop (game.py:279-328) driven by the scripted model of
    tests/golden/scripted.py — plain branch (planes + env scalars: get_multy(0) / [get_rate()], game.py:299,304) and
    MapNet branch (planes + prob_map plane, game.py:297) — replayed through this repo's Game.main_loop on the GPU:
    every model input, every action, the final board, positions, winner and history length are the reference's."""
    import sys
    from conftest import GOLDEN
    sys.path.insert(0, GOLDEN)
    from scripted import ScriptedModel
    from tron.game import Game, PositionPlayer
    from tron.player import ACPlayer
    from tron.util import pop_up
    g = load_golden("main_loop")
    branches = set()
    for i in range(int(g["n"])):
        pre = "c%d_" % i
        W, salt, n_steps, winner, hist, degree = (int(v) for v in g[pre + "scalars"])
        mode = None if str(g[pre + "mode"]) == "none" else str(g[pre + "mode"])
        slide = None if float(g[pre + "slide"]) == -999.0 else float(g[pre + "slide"])
        st = g[pre + "starts"]
        game = Game(W, W, [PositionPlayer(1, ACPlayer(), [int(st[0]), int(st[1])]),
                           PositionPlayer(2, ACPlayer(), [int(st[2]), int(st[3])])], mode, slide)
        game.weight = [int(v) for v in g[pre + "weight"]]
        game.degree = degree
        m1, m2 = ScriptedModel(salt=salt), ScriptedModel(salt=salt + 1)
        branch = str(g[pre + "branch"])
        branches.add(branch)
        if branch == "map":
            m1.wants_prob_plane = True                         # this repo's marker for the MapNet calling convention
        game.main_loop(m1, pop=pop_up, window=None, model2=m2)
        for who, m in (("p1", m1), ("p2", m2)):
            assert len(m.log) == n_steps, (i, who)
            assert np.array_equal([e["action"] for e in m.log], g[pre + who + "_action"]), (i, who)
            assert np.array_equal([e["channels"] for e in m.log], g[pre + who + "_channels"]), (i, who)
            assert np.array_equal(np.array([e["checksum"] for e in m.log]).reshape(n_steps, 3), g[pre + who + "_checksum"]), (i, who)
            assert np.array_equal(np.array([e["plane4"] for e in m.log]), g[pre + who + "_plane4"], equal_nan=True), (i, who)
            assert all(e["plane4_uniform"] for e in m.log)
            assert np.array_equal(np.array([e["env"] for e in m.log]).reshape(n_steps, -1), g[pre + who + "_env"]), (i, who)
        assert (0 if game.winner is None else game.winner) == winner and game.done, i
        assert np.array_equal(game._env.grid()[0].cpu().numpy(), g[pre + "grid"]), i
        pos = [game.pps[0].position[0], game.pps[0].position[1], game.pps[1].position[0], game.pps[1].position[1]]
        assert np.array_equal(pos, g[pre + "pos"]) and np.array_equal([int(pp.alive) for pp in game.pps], g[pre + "alive"]), i
        assert len(game.history) == hist, i
    assert branches == {"plain", "map"}


def test_device_replay_ring_and_sampling():
    from tron.vec import DeviceReplay, pop_up_planes
    cells, cap = 144, 1000
    rb = DeviceReplay(cap, cells, seed=11)
    rs = np.random.RandomState(0)
    vals = np.array([1, -1, -2, -3, 10, -10], np.int8)
    host = dict(s=np.zeros((cap, cells), np.int8), s2=np.zeros((cap, cells), np.int8), a=np.zeros(cap, np.int8),
                r=np.zeros(cap, np.float32), d=np.zeros(cap, np.int8))
    head = size = 0
    for n in (300, 450, 400, 77):                                   # wraps the ring twice
        s, s2 = vals[rs.randint(0, 6, (n, cells))], vals[rs.randint(0, 6, (n, cells))]
        a, r, d = rs.randint(0, 4, n).astype(np.int8), rs.randn(n).astype(np.float32), (rs.rand(n) < .3).astype(np.int8)
        rb.add(*(torch.from_numpy(x).cuda() for x in (s, a, r, s2, d)))
        idx = (head + np.arange(n)) % cap
        host["s"][idx], host["s2"][idx], host["a"][idx], host["r"][idx], host["d"][idx] = s, s2, a, r, d
        head, size = (head + n) % cap, min(size + n, cap)
        assert len(rb) == size
    counts = np.zeros(cap)
    for _ in range(200):
        st, a, r, s2, d = rb.sample(64, channels=4, plane4=5.0, side=12)
        idx = rb.last_indices(64).cpu().numpy()
        assert len(set(idx.tolist())) == 64 and idx.min() >= 0 and idx.max() < size      # random.sample: distinct
        counts[idx] += 1
        exp = pop_up_planes(torch.from_numpy(host["s"][idx].reshape(64, 12, 12)).cuda())
        assert torch.equal(st[:, :3], exp) and torch.all(st[:, 3] == 5.0)
        exp2 = pop_up_planes(torch.from_numpy(host["s2"][idx].reshape(64, 12, 12)).cuda())
        assert torch.equal(s2[:, :3], exp2)
        assert np.array_equal(a.cpu().numpy().ravel(), host["a"][idx].astype(np.int64))
        assert np.array_equal(r.cpu().numpy().ravel(), host["r"][idx])
        assert np.array_equal(d.cpu().numpy().ravel(), host["d"][idx].astype(np.float32))
    # uniformity: 12800 draws over 1000 slots, expected 12.8 each; chi-square far from pathological
    chi2 = ((counts - 12.8) ** 2 / 12.8).sum()
    assert 800 < chi2 < 1250, chi2
    from tron import _native as nat
    with pytest.raises(nat.TronNativeError):                         # random.sample raises when k > len(memory)
        rb.sample(2048, channels=3, side=12)
    rb2 = DeviceReplay(4096, cells, seed=12)
    s = torch.from_numpy(vals[rs.randint(0, 6, (3000, cells))]).cuda()
    z = torch.zeros(3000, device="cuda")
    rb2.add(s, z.to(torch.int8), z, s, z.to(torch.int8))
    big = rb2.sample(2048, channels=3, side=12)                      # any batch <= size: still distinct slots
    assert big[0].shape == (2048, 3, 12, 12)
    idx = rb2.last_indices(2048)
    assert int(idx.min()) >= 0 and int(idx.max()) < 3000 and idx.unique().numel() == 2048
    assert torch.equal(big[0], pop_up_planes(s[idx].reshape(2048, 12, 12)))


def test_dropin_agent_and_batched_trainer():
    import DDQN
    torch.manual_seed(0)
    agent = DDQN.Agent(10, 4, buffer_size=4096, batch_size=32, seed=5)
    from tron.util import make_game, pop_up
    game = make_game(True, True)

    def obs(p):
        planes = np.concatenate([pop_up(game.map().state_for_player(p)), game.prob_map()[None]], 0)
        return torch.from_numpy(planes[None]).float()
    agent.epsilon = 1.0
    s1 = obs(1)
    a1 = agent.action(s1)
    assert a1 in (0, 1, 2, 3)
    for k in range(140):                                             # learn() fires once len > batch and every 4th call
        agent.step(s1, k % 4, -1.0, s1, k % 7 == 0)
    assert len(agent.memory) == 140 and agent.steps >= 20
    assert torch.isfinite(agent.get_loss())
    out = DDQN.train(n_envs=512, width=10, steps=12, learn_every=2, batch_size=64, capacity=1 << 14, log_every=0)
    assert out["env_steps"] == 512 * 12 and out["transitions_pushed"] == 2 * 512 * 12
    assert out["learn_steps"] == 6 and out["games"] > 0
    assert len(out["brain"].memory) == min(2 * 512 * 12, 1 << 14)


def test_batched_dqn_train_cycle():
    """DQN.train (DQN.py:135-308) on the batched env: cycles of finished games, one smooth-L1 step per cycle on a sample of the
    device ring, epsilon x 0.999 per finished game, the reference's +100 / -25 / step-index rewards in the ring."""
    import torch
    import DQN
    torch.manual_seed(4)
    net = DQN.Net(in_channels=1, width=10)
    w0 = [p.detach().clone() for p in net.parameters()]
    logs = []
    out = DQN.train(model=net, n_envs=64, width=10, cycles=3, games_per_cycle=96, batch_size=64, capacity=4096, seed=9, log=logs.append)
    assert out["cycles"] == 3 and len(out["losses"]) == 3 and all(np.isfinite(out["losses"]))
    assert out["games"] >= 3 * 96 and out["draws"] + out["wins_p1"] + out["wins_p2"] == out["games"]
    assert abs(out["epsilon"] - DQN.EPSILON_START * DQN.DECAY_RATE ** out["games"]) < 1e-9          # far from the floor: one decay per game
    assert any((p.detach().cpu() - q).abs().max().item() > 0 for p, q in zip(out["model"].parameters(), w0))   # the net was trained
    assert [l["cycle"] for l in logs] == [0, 1, 2] and logs[-1]["games"] == out["games"]
    # the ring holds the reference's rewards: terminal +100 / -25 / 0, otherwise the step index
    from tron.vec import VecTron, DeviceReplay
    env = VecTron(32, 10, seed=2, obs_format="codes", reward="dqn")
    ring = DeviceReplay(1024, 144, seed=2)
    codes = env.reset().reshape(64, 12, 12)
    seen = set()
    for _ in range(12):
        ring.add_states(codes)
        obs, reward, done, winner = env.step(autoreset=False)
        r, d, w = reward.cpu().numpy(), done.cpu().numpy().astype(bool), winner.cpu().numpy()
        for i in range(32):
            if d[i]:
                want = {0: (0.0, 0.0), 1: (100.0, -25.0), 2: (-25.0, 100.0)}[int(w[i])]
                assert tuple(r[i]) == want
                seen.add(int(w[i]))
            else:
                assert r[i, 0] == r[i, 1] and r[i, 0] >= 0 and float(r[i, 0]).is_integer()
        env.reset(mask=done)
        codes = env.obs.reshape(64, 12, 12)
    assert seen


def test_batched_rating_sweep():
    """play.py:72-98 as one batch: 3 slide values x 400 fair/ice games between two random policies."""
    import play

    class Rand:
        def act(self, x, env=None):
            return torch.randint(0, 4, (x.shape[0],), device=x.device)

    torch.manual_seed(0)
    res = play.rating(Rand(), Rand(), n_games=400, slides=[0.0, 0.18, 0.36], width=10, verbose=False)
    assert [r["slide"] for r in res] == [0.0, 0.18, 0.36]
    for r in res:
        assert r["p1_win"] + r["p2_win"] + r["draw"] == 400
        assert 0.3 < r["p1_rate"] < 0.7           # symmetric random play
    # the "minimax rating" (ACKTR.py:408-421): a random policy against MinimaxPlayer(2, "voronoi") loses
    res = play.rating(Rand(), "minimax", n_games=300, slides=[0.0, 0.15], width=10, verbose=False)
    for r in res:
        assert r["p1_win"] + r["p2_win"] + r["draw"] == 300 and r["p1_rate"] < 0.2


def test_readme_encoding_variant():
    """README.md:67's wording of the observation (EMPTY 0 instead of the code's 1); un-oracled, see the docstring."""
    from tron.util import make_game, readme_encoding
    game = make_game(True, True)
    codes = game.map().state_for_player(1)
    r = readme_encoding(codes)
    assert set(np.unique(r).tolist()) <= {0, -1, 10, -10, -2, -3} and (r == 0).sum() == (codes == 1).sum()
    assert (r == 10).sum() == 1 and (r == -10).sum() == 1 and np.array_equal(r != 0, codes != 1)
    t = torch.as_tensor(codes).cuda()
    assert np.array_equal(readme_encoding(t).cpu().numpy(), r)


def test_rating_with_the_reference_player_pair():
    """play.py:53-61's seating — a MapNet as player 1 (planes + prob_map plane), a TestNet as player 2 (planes +
    [get_rate()]) — through the batched rating loop, both loaded by play.load_player."""
    import play
    torch.manual_seed(1)
    p1, p2 = play.load_player(None, "map"), play.load_player(None, "test")
    res = play.rating(p1, p2, n_games=128, slides=[0.0, 0.3], width=10, verbose=False)
    for r in res:
        assert r["p1_win"] + r["p2_win"] + r["draw"] == 128


def test_acktr_batched_trainer_runs():
    """ACKTR.train on VecTron: MapNet (planes4 from the kernel) with K-FAC, and Mulnet with A2C."""
    import ACKTR
    out = ACKTR.train(n_envs=64, width=10, model="map", reward="1", iterations=3, acktr=True, seed=3)
    assert out["env_steps"] == 3 * 5 * 64 and out["updates"] == 6 and out["games"] > 0
    assert all(np.isfinite(out["last_stats"]))
    assert out["brain"].optimizer.steps == 6
    out = ACKTR.train(n_envs=48, width=6, model="mul", reward="3", iterations=2, acktr=False, gamemode="temper", seed=4)
    assert out["env_steps"] == 2 * 5 * 48 and all(np.isfinite(out["last_stats"]))
    # ai_p2=False: player 2's moves come from the minimax search (ACKTR.py:286-287)
    out = ACKTR.train(n_envs=32, width=6, model="mul", reward="3", iterations=2, acktr=False, seed=5, ai_p2=False)
    assert out["env_steps"] == 2 * 5 * 32 and all(np.isfinite(out["last_stats"]))


def _tiles_from_codes(codes, player):
    """Invert Map.state_for_player(player) (map.py:67-84; slide tiles come back as bodies)."""
    own_body, own_head, foe_body, foe_head = (1, 2, 3, 4) if player == 1 else (3, 4, 1, 2)
    lut = {1: 0, -1: -1, -2: own_body, -3: foe_body, 10: own_head, -10: foe_head}
    return np.vectorize(lut.get, otypes=[np.int8])(codes)


def test_minimax_player_facade(monkeypatch):
    """MinimaxPlayer(2, mode).action(map, id) -> Direction, on boards recorded from the reference;
    random.getrandbits is pinned to the draw the reference's root consumed."""
    import random
    from golden.netgen import mm_stream
    from tron.map import Map
    from tron.minimax import MinimaxPlayer, Mode, Minimax
    from tron.player import Direction
    g = load_golden("minimax")
    k = "W10_"
    sel = np.nonzero(g[k + "depth"] == 2)[0][:60]
    for i in sel:
        pid = int(g[k + "player"][i])
        m = Map.from_codes(10, _tiles_from_codes(g[k + "codes"][i], pid))
        assert np.array_equal(m.state_for_player(pid), g[k + "codes"][i])
        draw = int(mm_stream(int(g[k + "seed"][i]), 96)[int(g[k + "draws"][i]) - 1])
        monkeypatch.setattr(random, "getrandbits", lambda n, d=draw: d)
        mode = Mode.DISTWALL if int(g[k + "mode"][i]) == 1 else "voronoi"
        d = MinimaxPlayer(2, mode).action(m, pid)
        assert isinstance(d, Direction) and d.value == int(g[k + "move"][i]), int(i)
        assert Minimax(2, mode).get_move(g[k + "codes"][i].T) == int(g[k + "move"][i])
    with pytest.raises(Exception):
        MinimaxPlayer(3, "voronoi").action(m, pid)          # depths the reference never builds


def test_make_game_against_minimax():
    """make_game(True, False): player 2 is MinimaxPlayer(2, "voronoi") and ignores the action it is
    handed (game.py:179-181); each of its moves is one of the oracle's best moves for the map
    before the step."""
    import oracle
    from tron.util import make_game
    from tron.minimax import MinimaxPlayer
    rng = np.random.default_rng(3)
    games = moves = 0
    while games < 10:
        game = make_game(True, False)
        assert isinstance(game.pps[1].player, MinimaxPlayer) and not isinstance(game.pps[0].player, MinimaxPlayer)
        while not game.done:
            before = game.map().state_for_player(2)
            _, values, expanded, _ = oracle.minimax_move(before, 2, 0, np.zeros(8, np.uint32))
            game.step(int(rng.integers(4)), None)
            d = game.pps[1].player.direction.value - 1
            if expanded.any():
                best = values[expanded].max()
                assert expanded[d] and values[d] == best, (games, moves)
            moves += 1
        games += 1
    assert moves > 15
