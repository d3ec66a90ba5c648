"""Game — the reference's scalar game object as a view onto ONE env of the HIP path.

Same constructor, attributes and methods as tron/game.py (Game: game.py:70-328,
PositionPlayer: :36-58, HistoryElement: :61-65).  Every rule runs in csrc/tron_env.hip;
this class uploads the two actions, launches the one-env kernel and mirrors the results
back into the Python objects the reference's trainers read.  For throughput use
tron.vec.VecTron — this facade costs a launch and a few small copies per step."""
import random
from time import sleep

import numpy as np
import torch

from .map import Map, Tile
from .player import ACPlayer, Direction
from .vec import VecTron
from . import _native as nat

__all__ = ["Game", "PositionPlayer", "HistoryElement", "Winner"]


class Winner:                                     # game.py:31-33 (unused there too)
    PLAYER_ONE = 1
    PLAYER_TWO = 2


class PositionPlayer:                             # game.py:36-58
    def __init__(self, id, player, position):
        self.id = id
        self.player = player
        self.position = position
        self.alive = True

    def body(self):
        return Tile.PLAYER_ONE_BODY if self.id == 1 else Tile.PLAYER_TWO_BODY

    def slide(self):
        return Tile.PLAYER_ONE_slide if self.id == 1 else Tile.PLAYER_TWO_slide

    def head(self):
        return Tile.PLAYER_ONE_HEAD if self.id == 1 else Tile.PLAYER_TWO_HEAD


class HistoryElement:                             # game.py:61-65
    def __init__(self, mmap, player_one_direction, player_two_direction):
        self.map = mmap
        self.player_one_direction = player_one_direction
        self.player_two_direction = player_two_direction


class Game:
    def __init__(self, width, height, pps, mode=None, slide_pram=None, _fair_start=None):
        if width != height:
            raise ValueError("boards are square (map.py:48 / util.py:16-17 are only right for w == h)")
        if len(pps) != 2:
            raise ValueError("two players")
        self.width = width
        self.height = height
        self.pps = pps
        self.winner = None
        self.next_p1 = []
        self.next_p2 = []
        self.done = False
        self.mode = mode
        # the reference draws weight/degree from the unseeded `random` module (game.py:83,87);
        # here they come from the env's Philox stream, keyed by a seed taken from `random`
        self._env = VecTron(1, width, mode=mode, fair=bool(_fair_start), seed=random.getrandbits(32), rank=0,
                            obs_format="codes", reward="ddqn")
        from config import slide as default_slide
        self._slide = default_slide if slide_pram is None else slide_pram        # game.py:88
        self._env.set_slide(float(self._slide))
        if _fair_start is None:
            start = torch.tensor([[pps[0].position[0], pps[0].position[1], pps[1].position[0], pps[1].position[1]]],
                                 dtype=torch.int8)
            self._env.reset(start_pos=start)
        else:                                     # util.make_game: positions drawn on the device
            self._env.reset()
        self._pull()
        self.history = [HistoryElement(self._snapshot(), None, None)]             # game.py:76,90-91

    # ---- device <-> python mirrors --------------------------------------------------
    def _snapshot(self):
        return Map.from_codes(self.width, self._env.grid()[0].cpu().numpy())

    def _pull(self):
        st = {k: v.cpu().numpy() for k, v in self._env.state().items()}
        pos = st["pos"][0]
        for i, pp in enumerate(self.pps):
            pp.position = (int(pos[2 * i]), int(pos[2 * i + 1]))
            pp.alive = bool(st["alive"][0, i])
            d = int(st["dir"][0, i])
            if d and hasattr(pp.player, "__dict__"):
                pp.player.direction = Direction(d)
        self._weight = [int(st["weight"][0, 0]), int(st["weight"][0, 1])]
        self._degree = int(st["degree"][0])
        self.done = bool(st["done"][0])
        w = int(st["winner"][0])
        self.winner = None if w == 0 else w
        return st

    # Game.weight / .degree / .slide are plain attributes in the reference; assignments reach the device
    @property
    def weight(self):
        return self._weight

    @weight.setter
    def weight(self, v):
        self._weight = [int(v[0]), int(v[1])]
        self._env.set_weight_degree(weight=torch.tensor([self._weight], dtype=torch.int16))

    @property
    def degree(self):
        return self._degree

    @degree.setter
    def degree(self, v):
        self._degree = int(v)
        self._env.set_weight_degree(degree=torch.tensor([self._degree], dtype=torch.int16))

    @property
    def slide(self):
        return self._slide

    @slide.setter
    def slide(self, v):
        self._slide = v
        self._env.set_slide(float(v))

    # ---- reference API -----------------------------------------------------------------
    def map(self):                                # game.py:93-94
        return self.history[-1].map.clone()

    def get_rate(self, player_num=None):          # game.py:96-102 (accessor; the kernel has its own)
        if player_num is None:
            return -((self.degree - 30) * 0.6) / 100
        return (-((self.degree - 30) * 0.6) / 100) - ((70 - self.get_weight(player_num)) / 100)

    def get_degree(self):                         # game.py:105-108
        return float(self.degree)

    def get_degree_silde(self):                   # game.py:110-112
        return float((-self.slide * 100) * (10 / 6) + 30)

    def change_degree(self):                      # game.py:114-122 (never called by the reference's step)
        if random.random() > 0.5:
            self.degree = min(30, self.degree + random.randint(0, 3))
        else:
            self.degree = max(-30, self.degree - random.randint(1, 5))

    def prob_map(self):                           # game.py:124-132 — sized by this game, not by config
        return np.full((self.width + 2, self.height + 2), self.get_degree_silde())

    def get_weight(self, player_num):             # game.py:133-135
        return self.weight[player_num]

    def get_multy(self, player_num):              # game.py:137-139
        return [self.get_degree(), self.get_weight(player_num)]

    def degree_map(self):                         # game.py:140-147
        return np.full((self.width + 2, self.height + 2), self.get_degree())

    def next_frame(self, action_p1, action_p2, window=None):
        """One move of both players on the device (game.py:149-252).  Returns True."""
        if self.done:
            raise RuntimeError("stepping a finished game is undefined in the reference; make a new Game")
        a = torch.tensor([[int(action_p1 or 0) & 3, int(action_p2 or 0) & 3]], dtype=torch.int8)
        for i, pp in enumerate(self.pps):         # game.py:179-181: a non-AC player picks its own move
            if hasattr(pp.player, "minimax"):
                from .minimax import _kernel_mode
                a = a.to(self._env.device)
                a[:, i] = self._env.minimax_actions(i + 1, _kernel_mode(pp.player.mode))
        obs, _, _, _ = self._env.step(a, autoreset=False)
        o = obs[0].cpu().numpy().astype(np.int64)
        self._pull()
        self.history[-1].player_one_direction = self.pps[0].player.direction      # game.py:200-201
        self.history[-1].player_two_direction = self.pps[1].player.direction
        self.history.append(HistoryElement(self._snapshot(), None, None))         # game.py:230
        self.next_p1, self.next_p2 = o[0], o[1]                                   # game.py:231-232
        return True

    def step(self, action_p1, action_p2):         # game.py:254-277
        self.next_frame(action_p1, action_p2)
        return self.next_p1, self.next_p2, self.done

    def main_loop(self, model, pop=None, window=None, model2=None):
        """Self-play to termination with `model.act` choosing the moves (game.py:279-328).
        A model with `wants_prob_plane = True` gets the 4-plane input of the MapNet branch
        (game.py:297); others get (obs, env_scalars) like game.py:299,304."""
        from config import device
        if pop is None:
            from .util import pop_up as pop       # the reference passes pop=None and crashes
        if window:
            window.render_map(self.map())
        if not model2:
            model2 = model
        while True:
            if window:
                sleep(0.3)
            m = self.map()
            with torch.no_grad():
                acts = []
                for pid, mdl in ((1, model), (2, model2)):
                    planes = torch.tensor(pop(m.state_for_player(pid)))
                    if getattr(mdl, "wants_prob_plane", False):
                        x = torch.cat([planes, torch.tensor(self.prob_map()).unsqueeze(0)], 0).unsqueeze(0).float()
                        acts.append(mdl.act(x))
                    else:
                        env = self.get_multy(0) if pid == 1 else [self.get_rate()]
                        acts.append(mdl.act(planes.unsqueeze(0).float(), torch.tensor([env]).to(device)))
            a1, a2 = (int(torch.as_tensor(a).reshape(-1)[0]) for a in acts)
            self.next_frame(a1, a2, window)
            if self.done:                         # game.py:315-325 — same winner rule as step()
                break
            if window:
                window.render_map(self.map())
