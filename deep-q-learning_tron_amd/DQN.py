"""Plain DQN pieces of the reference's DQN.py: `Ai(epsilon).action(map, id) -> Direction`
(DQN.py:39-75), the `ReplayMemory` ring (:81-132) and the smooth-L1 learn step (:262-292).
The reference's own `train()` cannot run at HEAD; `learn_step` is its
loss restated for a network that takes the observation it is actually given, and `train()` here is its
loop (DQN.py:135-308: cycles of 20 self-play games, then ONE smooth-L1 step on a sample of the ring)
on the batched env and the HBM ring."""
import random
from collections import namedtuple

import numpy as np
import torch
import torch.nn.functional as F

from tron.player import Player, Direction
from Net.DQNNet import Net

BATCH_SIZE = 128             # DQN.py:19-36
GAMMA = 0.9
EPSILON_START = 1
ESPILON_END = 0.003
DECAY_RATE = 0.999
MAP_WIDTH = 10
MAP_HEIGHT = 10
MEM_CAPACITY = 10000
GAME_CYCLE = 20
DISPLAY_CYCLE = GAME_CYCLE
device = 'cuda' if torch.cuda.is_available() else 'cpu'

Transition = namedtuple('Transition', ('old_state', 'action', 'new_state', 'reward', 'terminal'))


class Ai(Player):
    """DQN.py:39-75 — argmax of the net on the (1,1,S,S) code plane, epsilon-random otherwise."""

    def __init__(self, epsilon=0, width=MAP_WIDTH):
        super(Ai, self).__init__()
        self.net = Net(in_channels=1, width=width).to(device)
        self.epsilon = epsilon

    def action(self, map, id):
        game_map = map.state_for_player(id)
        inp = torch.from_numpy(np.reshape(game_map, (1, 1, game_map.shape[0], game_map.shape[1]))).float()
        with torch.no_grad():
            output = self.net(inp)
        next_action = int(torch.max(output.data, 1)[1].cpu().numpy()[0]) + 1
        if random.random() <= self.epsilon:
            next_action = random.randint(1, 4)
        return Direction(next_action)


class ReplayMemory(object):
    """DQN.py:81-132 — a host ring of Transition tuples (capacity 1e4); kept for surface parity.
    The batched path uses tron.vec.DeviceReplay instead."""

    def __init__(self, capacity):
        self.capacity = capacity
        self.memory = []
        self.position = 0

    def push(self, *args):
        if len(self.memory) < self.capacity:
            self.memory.append(None)
        self.memory[self.position] = Transition(*args)
        self.position = (self.position + 1) % self.capacity

    def sample(self, batch_size):
        return random.sample(self.memory, batch_size)

    def __len__(self):
        return len(self.memory)


def learn_step(model, optimizer, old_states, actions, new_states, rewards, terminals, gamma=GAMMA):
    """DQN.py:262-292: y = r if terminal else r + gamma max_a Q(s', a); smooth-L1; one optimiser step."""
    pred = model(old_states).gather(1, actions.long()).sum(dim=1)
    with torch.no_grad():
        nxt = model(new_states).max(1)[0]
        target = rewards.reshape(-1) + gamma * nxt * (1.0 - terminals.reshape(-1).float())
    model.zero_grad()
    loss = F.smooth_l1_loss(pred, target)
    loss.backward()
    optimizer.step()
    return loss.detach()


def observation_plane(codes):
    """The network input of the reference's DQN: the observation codes themselves as ONE float plane (DQN.py:54-57,186-191:
    `state_for_player` reshaped to (1, 1, S, S)), int8 [n, S, S] -> f32 [n, 1, S, S]."""
    return codes.to(torch.float32).unsqueeze(1)


def train(model=None, n_envs=1024, width=MAP_WIDTH, cycles=10, games_per_cycle=None, batch_size=BATCH_SIZE, capacity=MEM_CAPACITY,
          seed=0x5EED, save_path=None, log=None):
    """The reference's DQN.train (DQN.py:135-308) on N parallel self-play envs.

    One cycle = GAME_CYCLE (20) finished games PER ENV SLOT on average — `games_per_cycle` = 20 x n_envs finished games by
    default (the reference plays its 20 games one after the other; here the N slots play side by side with auto-restart) —
    then ONE optimiser step: sample min(len(memory), batch) transitions, y = r if terminal else r + gamma max_a Q(s', a),
    smooth-L1, Adam (DQN.py:258-292).  Both players are the same net acting epsilon-greedily on its own observation plane
    (DQN.py:51-75); transitions are (s, a, s', r, terminal) per player with the reference's rewards — the step index for a
    non-terminal move, +100 / -25 / 0 at the end (DQN.py:224-241: tron.vec's "dqn" table) — pushed online into the device
    ring instead of being replayed out of `game.history` after the game (H1).  Epsilon decays by 0.999 per
    finished game down to 0.003 (DQN.py:252-255), applied per step for the games that step finished.
    Returns a dict of counters; `model` defaults to a fresh `Net(1, width)`."""
    import time
    from tron.vec import VecTron, DeviceReplay
    dev = torch.device("cuda")
    if model is None:
        torch.manual_seed(seed)
        model = Net(in_channels=1, width=width)
    model = model.to(dev)
    optimizer = torch.optim.Adam(model.parameters())                       # DQN.py:139
    S = width + 2
    env = VecTron(n_envs, width, mode=None, seed=seed, obs_format="codes", reward="dqn")
    memory = DeviceReplay(capacity, S * S, seed=seed)
    games_per_cycle = GAME_CYCLE * n_envs if games_per_cycle is None else int(games_per_cycle)
    epsilon, games, moves, losses = float(EPSILON_START), 0, 0, []
    wins = torch.zeros(3, dtype=torch.int64, device=dev)                    # draws, player 1, player 2
    gen = torch.Generator(device=dev).manual_seed(seed)
    codes = env.reset().reshape(2 * n_envs, S, S)
    t0 = time.perf_counter()
    for cycle in range(cycles):
        cycle_games = 0
        while cycle_games < games_per_cycle:
            with torch.no_grad():
                model.eval()
                greedy = model(observation_plane(codes)).argmax(1).to(torch.int8)
                model.train()
            explore = torch.rand(2 * n_envs, device=dev, generator=gen) <= epsilon
            actions = torch.where(explore, torch.randint(0, 4, (2 * n_envs,), device=dev, generator=gen, dtype=torch.int8), greedy)
            memory.add_states(codes)                                        # s, before the step overwrites the observation buffer
            obs, reward, done, winner = env.step(actions.reshape(n_envs, 2), autoreset=False)
            memory.add(None, actions, reward.reshape(-1), obs.reshape(2 * n_envs, S, S), done.repeat_interleave(2))
            finished = int(done.sum())                                      # (one host read per step: epsilon's schedule is per game)
            if finished:
                wins += torch.bincount(winner[done.bool()].long(), minlength=3)
                env.reset(mask=done)
                for _ in range(finished):                                   # DQN.py:252-255 (the rule is monotone: once it stops it stays stopped)
                    if epsilon * DECAY_RATE <= ESPILON_END:
                        break
                    epsilon *= DECAY_RATE
            codes = env.obs.reshape(2 * n_envs, S, S)
            moves += n_envs
            cycle_games += finished
        games += cycle_games
        n = min(len(memory), batch_size)                                    # DQN.py:258
        st, a, r, s2, d = memory.sample_codes(n, side=S)
        loss = learn_step(model, optimizer, observation_plane(st), a, observation_plane(s2), r, d)
        losses.append(float(loss))
        if save_path:
            torch.save(model.state_dict(), save_path)                       # DQN.py:295
        if log:
            log(dict(cycle=cycle, games=games, loss=losses[-1], epsilon=epsilon))
    torch.cuda.synchronize()
    w = wins.cpu().tolist()
    env.close()
    memory.close()
    return dict(model=model, cycles=cycles, games=games, env_steps=moves, losses=losses, epsilon=epsilon, seconds=time.perf_counter() - t0,
                draws=w[0], wins_p1=w[1], wins_p2=w[2])
