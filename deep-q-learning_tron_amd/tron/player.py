"""Player kinematics — same surface as the reference's tron/player.py.

The move itself is executed by the HIP kernel (csrc/tron_env.hip, lane_move); these classes
keep the reference's names so `PositionPlayer(1, ACPlayer(), [x, y])` and the trainers'
`Direction` arithmetic work unchanged.  player.py:4-8 (Direction), :11-41 (Player), :95-132
(ACPlayer).  KeyboardPlayer is pygame UI and out of scope."""
from enum import Enum


class Direction(Enum):       # player.py:4-8
    UP = 1
    RIGHT = 2
    DOWN = 3
    LEFT = 4


# action index 0..3 -> (d_row, d_col): UP, RIGHT, DOWN, LEFT on (row, col)  (player.py:107-132)
DELTAS = ((-1, 0), (0, 1), (1, 0), (0, -1))


class Player(object):
    """Abstract player (player.py:11-41): the hooks the reference's agents override."""

    def __init__(self):
        self.direction = None

    def find_file(self, name):
        pass

    def next_position(self, current_position, direction):
        pass

    def get_direction(self, current_position, direction):
        pass

    def next_position_and_direction(self, current_position, action):
        pass

    def action(self, map, id):
        pass

    def step(self, state, action, reward, next_step, done):
        pass

    def learn(self, experiences, gamma):
        pass

    def soft_update(self, local_model, target_model, tau):
        pass

    def manage_event(self, event):
        pass


class Mode(Enum):            # player.py:45-47
    ARROWS = 1
    ZQSD = 2


class ACPlayer(Player):
    """The env-driven player: the action index comes from the caller (player.py:95-132)."""

    def get_direction(self, next_action):
        return Direction(int(next_action) + 1)

    def next_position(self, current_position, direction):
        dr, dc = DELTAS[direction.value - 1]
        return current_position[0] + dr, current_position[1] + dc

    def next_position_and_direction(self, current_position, action):
        direction = self.get_direction(action)
        return self.next_position(current_position, direction), direction


class KeyboardPlayer(Player):
    """Placeholder for the pygame keyboard player (player.py:50-92): holds a direction only."""

    def __init__(self, initial_direction, mode=Mode.ARROWS):
        super().__init__()
        self.direction = initial_direction
        self.mode = mode

    def action(self, map, id):
        return self.direction
