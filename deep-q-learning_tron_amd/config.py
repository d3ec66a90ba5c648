"""Constants of the reference's config.py (config.py:1-41), same names so `from config import *`
keeps working.  Values are the reference's; grid size is a real parameter everywhere else."""
import os

# MIOpen's default exhaustive "find" costs tens of seconds for every new (batch, shape) the CNN sees;
# the trainers change batch sizes freely, so use the heuristic immediate mode unless the user chose one.
os.environ.setdefault("MIOPEN_FIND_MODE", "FAST")

import torch  # noqa: E402

device = 'cuda' if torch.cuda.is_available() else 'cpu'   # config.py:3

GAMMA = 0.9             # config.py:5
BATCH_SIZE = 64         # config.py:7 (DDQN)

lr = 3e-3               # config.py:10-12 (RMSprop, unused under acktr)
eps = 1e-5
alpha = 0.99

NUM_PROCESSES = 16      # config.py:14  envs stepped per iteration by ACKTR.py
NUM_ADVANCED_STEP = 5   # config.py:15

value_loss_coef = 0.5   # config.py:18-21
entropy_coef = 0.01
policy_loss_coef = 1
max_grad_norm = 0.5

MAP_WIDTH = 10          # config.py:23-24
MAP_HEIGHT = 10

SHOW_ITER = 20          # config.py:26
PLAY_WITH_MINIMAX = 200  # config.py:28

slide = 0.15            # config.py:32
GAME_MODE = "temper"    # config.py:34

reward_cons1 = [10, -10]      # config.py:37-41  (win, lose)
reward_cons2 = [10, -20]
reward_cons3 = [20.0, -10.0]
