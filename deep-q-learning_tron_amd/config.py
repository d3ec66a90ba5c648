"""Hyper-parameters under the names the reference's trainers star-import (`from config import *`,
config.py:1-41 there).  Kept as one table so the values, their meaning and the line they come
from sit together; grid size is a real parameter everywhere else in this package."""
import os

# MIOpen's default exhaustive "find" costs tens of seconds for every new (batch, shape) the CNN sees;
# the trainers change batch sizes freely, so use the heuristic immediate mode unless the user chose one.
os.environ.setdefault("MIOPEN_FIND_MODE", "FAST")
# In that mode MIOpen ranks solvers without a workspace, so every 3x3 weight gradient falls back to
# the fp32 Winograd kernel: 9.9 ms per call at batch 4096 x 12x12, 65 % of the DDQN trainer's GPU time
# (rocprofv3, round 1).  Without the Winograd family the implicit-GEMM kernels are picked instead:
# 44.6 K -> 118 K learned transitions/s.  (MIOPEN_FIND_MODE=NORMAL reaches 137 K after ~50 s of search.)
os.environ.setdefault("MIOPEN_DEBUG_CONV_WINOGRAD", "0")

import torch  # noqa: E402

_TABLE = (
    # name                value        reference line / meaning
    ("GAMMA",             0.9,         "config.py:5   discount"),
    ("BATCH_SIZE",        64,          "config.py:7   DDQN learn batch"),
    ("lr",                3e-3,        "config.py:10  RMSprop (A2C) learning rate"),
    ("eps",               1e-5,        "config.py:11  RMSprop epsilon"),
    ("alpha",             0.99,        "config.py:12  RMSprop smoothing"),
    ("NUM_PROCESSES",     16,          "config.py:14  envs stepped per ACKTR iteration"),
    ("NUM_ADVANCED_STEP", 5,           "config.py:15  n-step return length"),
    ("value_loss_coef",   0.5,         "config.py:18"),
    ("entropy_coef",      0.01,        "config.py:19"),
    ("policy_loss_coef",  1,           "config.py:20"),
    ("max_grad_norm",     0.5,         "config.py:21  (unused by the reference's update)"),
    ("MAP_WIDTH",         10,          "config.py:23"),
    ("MAP_HEIGHT",        10,          "config.py:24"),
    ("SHOW_ITER",         20,          "config.py:26  logging period"),
    ("PLAY_WITH_MINIMAX", 200,         "config.py:28  rating games"),
    ("slide",             0.15,        "config.py:32  default Game.slide"),
    ("GAME_MODE",         "temper",    "config.py:34  default ACKTR game mode"),
    ("reward_cons1",      [10, -10],   "config.py:37  (win, lose)"),
    ("reward_cons2",      [10, -20],   "config.py:39"),
    ("reward_cons3",      [20.0, -10.0], "config.py:41"),
)
globals().update({name: value for name, value, _ in _TABLE})
device = 'cuda' if torch.cuda.is_available() else 'cpu'      # config.py:3

__all__ = [name for name, _, _ in _TABLE] + ["device"]
