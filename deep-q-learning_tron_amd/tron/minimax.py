"""Minimax/Voronoi opponent — same surface as the reference's tron/minimax.py: `MinimaxPlayer(depth,
mode).action(map, id) -> Direction`, `Minimax(depth, mode).get_move(game_map)`, `Mode`.

The search itself (minimax.py:57-278: depth-2 tree over my moves and the opponent's replies,
leaves scored by Voronoi territory from two flood fills, or by wall distances) runs in
csrc/tron_minimax.hip behind `tron_minimax_codes` / `tron_minimax_actions`; one board here, N
boards at once through `VecTron.minimax_actions`.  The reference only ever builds
`MinimaxPlayer(2, "voronoi")` (util.py:82-83, ACKTR.py:13); other depths raise
TRON_ERR_UNSUPPORTED.  random.choice among equally good moves / random.randint for a boxed-in
head draw 32 bits from Python's `random`, like the reference's unseeded calls (minimax.py:238,270).
"""
import random
from enum import Enum

import numpy as np
import torch

from .player import Player, Direction, DELTAS
from .vec import minimax_codes


class Mode(Enum):            # minimax.py:281-284
    DISTWALL = 1
    VORNOI = 2


def _kernel_mode(mode):
    # minimax.py:228: anything that is not Mode.DISTWALL (e.g. the string "voronoi") scores by Voronoi
    return "distwall" if mode == Mode.DISTWALL else "voronoi"


class Minimax(object):
    def __init__(self, depth, mode):
        self.depth = depth
        self.mode = mode

    def get_move(self, game_map):
        """game_map: the TRANSPOSED observation codes, as MinimaxPlayer.action passes them
        (minimax.py:287).  Returns the move 1..4 = UP, RIGHT, DOWN, LEFT."""
        codes = torch.as_tensor(np.ascontiguousarray(np.asarray(game_map).T)).to(torch.int8).cuda()
        draw = torch.tensor([random.getrandbits(32)], dtype=torch.int64, device=codes.device)
        act, _, _ = minimax_codes(codes[None], draw, _kernel_mode(self.mode), self.depth)
        a = int(act[0])
        if a < 0:
            raise ValueError("minimax needs one +10 and one -10 head inside the border")
        return a + 1

    def __str__(self):
        return "Minimax"


class MinimaxPlayer(Player):
    def __init__(self, depth, mode=Mode.VORNOI):
        super(MinimaxPlayer, self).__init__()
        self.mode = mode
        self.depth = depth
        self.minimax = Minimax(depth, mode)
        self.direction = None

    def initialize_minimax(self):             # minimax.py:281-282
        self.minimax = Minimax(self.depth, self.mode)

    def action(self, map, id):                # minimax.py:284-297
        self.initialize_minimax()
        game_map = map.state_for_player(id).T
        return Direction(self.minimax.get_move(game_map))

    def next_position_and_direction(self, current_position, id, map, action=None):   # minimax.py:299-306
        direction = action if action is not None else self.action(map, id)
        return self.next_position(current_position, direction), direction

    def next_position(self, current_position, direction):                            # minimax.py:308-316
        dr, dc = DELTAS[direction.value - 1]
        return current_position[0] + dr, current_position[1] + dc
