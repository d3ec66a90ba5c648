"""A deterministic stand-in for a policy network, shared by the fixture generator (which drives the REFERENCE's
Game.main_loop with it) and the tests (which drive this repo's Game.main_loop with it).  It decides from the
observation planes alone — pop_up's (wall, my, enemy) planes, util.py:11-37 — and records what it was called with,
so the fixture pins both the inputs main_loop builds (game.py:294-304) and the game that results."""
import numpy as np
import torch

# action -> (d_row, d_col): UP, RIGHT, DOWN, LEFT (player.py:107-132)
_DELTA = ((-1, 0), (0, 1), (1, 0), (0, -1))


class ScriptedModel:
    """act(x, env=None): the first heading, in an order rotated by the call count and by `salt`, whose target cell
    is free in all three planes; heading `salt % 4` when none is (a losing move — games end).  x: [1, 3|4, S, S]."""

    def __init__(self, salt=0, log=None):
        self.salt = int(salt)
        self.calls = 0
        self.log = [] if log is None else log

    def act(self, x, env=None):
        x = torch.as_tensor(x).detach().cpu().double().numpy()
        assert x.ndim == 4 and x.shape[0] == 1 and x.shape[1] in (3, 4)
        planes = x[0]
        head = np.argwhere(planes[1] == 10.0)
        assert len(head) == 1, "exactly one own head in the `my` plane"
        r, c = int(head[0][0]), int(head[0][1])
        occupied = (planes[0] + planes[1] + planes[2]) != 0
        action = self.salt % 4
        for k in range(4):
            a = (self.calls + self.salt + k) % 4
            rr, cc = r + _DELTA[a][0], c + _DELTA[a][1]
            if not occupied[rr, cc]:
                action = a
                break
        env_vec = [] if env is None else [float(v) for v in torch.as_tensor(env).detach().cpu().double().reshape(-1)]
        # what the model saw: a position-weighted checksum of every plane, the 4th plane's value, the env scalars
        w = np.arange(planes[0].size, dtype=np.float64).reshape(planes[0].shape) + 1.0
        self.log.append(dict(call=self.calls, channels=int(x.shape[1]), checksum=[float((p * w).sum()) for p in planes[:3]],
                             plane4=float(planes[3, 0, 0]) if x.shape[1] == 4 else float("nan"),
                             plane4_uniform=bool(x.shape[1] < 4 or np.all(planes[3] == planes[3, 0, 0])),
                             env=env_vec, action=action))
        self.calls += 1
        return torch.tensor([[action]])
