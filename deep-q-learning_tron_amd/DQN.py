"""Plain DQN pieces of the reference's DQN.py: `Ai(epsilon).action(map, id) -> Direction`
(DQN.py:39-75), the `ReplayMemory` ring (:81-132) and the smooth-L1 learn step (:262-292).
The reference's own `train()` cannot run at HEAD; `learn_step` is its
loss restated for a network that takes the observation it is actually given."""
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
