"""Env helpers — same surface as the reference's tron/util.py: pop_up, prob_map, make_game,
get_reward (util.py:11-94).  pop_up runs on the GPU (tron_pop_up); make_game draws its start
positions on the device with the reference's rule (P1-only redraw on a clash, "fair" window)."""
import numpy as np
import torch

from config import *          # noqa: F401,F403  (the reference star-imports config here too)
from .game import *           # noqa: F401,F403
from .game import Game, PositionPlayer
from .player import ACPlayer, KeyboardPlayer  # noqa: F401
from .minimax import MinimaxPlayer
from .vec import pop_up_planes


def pop_up(map):
    """(S, S) observation codes -> (3, S, S) float64 planes (wall, my, enemy); heads carry 10
    (util.py:11-37)."""
    codes = torch.as_tensor(np.ascontiguousarray(map)).to(torch.int8).cuda()
    return pop_up_planes(codes[None])[0].cpu().numpy().astype(np.float64)


def readme_encoding(codes):
    """The observation as the reference's README.md:67 words it — "0 for empty space, -1 for a wall, 10 for the
    player head and -10 for the enemy's head" — from Map.state_for_player's codes, whose only difference is EMPTY = 1
    (map.py:68-69).  No reference code produces this variant, so it is pinned by nothing but that
    sentence: bodies keep the code's -2 / -3.  Works on numpy arrays and torch tensors."""
    if torch.is_tensor(codes):
        return torch.where(codes == 1, torch.zeros_like(codes), codes)
    codes = np.asarray(codes)
    return np.where(codes == 1, 0, codes).astype(codes.dtype)


def prob_map(x, width=None, height=None):         # util.py:38-45; size defaults to config like the reference
    import config
    return np.full(((width or config.MAP_WIDTH) + 2, (height or config.MAP_HEIGHT) + 2), float(x))


def make_game(p1, p2, mode=None, gamemode=None, slide_pram=None, width=None):
    """util.py:46-84.  p1/p2 False puts MinimaxPlayer(2, "voronoi") in that seat (util.py:82-83)."""
    import config
    w = width or config.MAP_WIDTH
    pps = [PositionPlayer(1, ACPlayer() if p1 else MinimaxPlayer(2, "voronoi"), [0, 0]),
           PositionPlayer(2, ACPlayer() if p2 else MinimaxPlayer(2, "voronoi"), [0, 0])]
    return Game(w, w, pps, gamemode, slide_pram, _fair_start=(mode == "fair"))


def get_reward(game, constants):                  # util.py:87-94
    if game.winner is None:
        return 0, 0
    elif game.winner == 1:
        return constants[0], constants[1]
    else:
        return constants[1], constants[0]
