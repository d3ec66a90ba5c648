"""Grid and per-player observation — same surface as the reference's tron/map.py.

A `Map` is a host snapshot of one board: `(w+2) x (h+2)` Tile codes with a WALL border,
interior cell (i, j) at storage [i+1][j+1] (map.py:45-48, 86-92).  The observation encode
`state_for_player` (map.py:67-84) runs on the GPU through tron_encode_codes; there is no
host implementation of it in this package."""
from enum import Enum

import numpy as np


def is_on_border(i, j, w, h):                    # map.py:5-6
    return i == 0 or i == w - 1 or j == 0 or j == h - 1


class Tile(Enum):                                # map.py:9-17
    EMPTY = 0
    WALL = -1
    PLAYER_ONE_BODY = 1
    PLAYER_ONE_HEAD = 2
    PLAYER_TWO_BODY = 3
    PLAYER_TWO_HEAD = 4
    PLAYER_ONE_slide = 5
    PLAYER_TWO_slide = 6

    def color(self):                             # map.py:21-41 (used by the pygame window only)
        return _TILE_RGB.get(self)


_TILE_RGB = {
    Tile.EMPTY: (0, 0, 0), Tile.WALL: (255, 255, 255),
    Tile.PLAYER_ONE_BODY: (0, 17, 128), Tile.PLAYER_ONE_HEAD: (0, 34, 255), Tile.PLAYER_ONE_slide: (0, 180, 250),
    Tile.PLAYER_TWO_BODY: (128, 17, 0), Tile.PLAYER_TWO_HEAD: (255, 34, 0), Tile.PLAYER_TWO_slide: (250, 100, 0),
}
_TILE_OF = {t.value: t for t in Tile}


class Map:
    def __init__(self, w, h, empty, wall):
        if w != h:
            raise ValueError("boards are square: the reference's border test is only right for w == h (map.py:48)")
        self.width = w
        self.height = h
        e = empty.value if isinstance(empty, Tile) else int(empty)
        wl = wall.value if isinstance(wall, Tile) else int(wall)
        self._data = np.full((w + 2, h + 2), e, dtype=np.int8)
        self._data[0, :] = self._data[-1, :] = self._data[:, 0] = self._data[:, -1] = wl

    @classmethod
    def from_codes(cls, w, codes):
        m = cls(w, w, 0, 0)
        m._data = np.array(codes, dtype=np.int8).reshape(w + 2, w + 2)
        return m

    def clone(self):                             # map.py:50-53
        return Map.from_codes(self.width, self._data.copy())

    def array(self):                             # map.py:60-61 — here an int8 image of Tile values
        return self._data

    def clone_array(self):                       # map.py:63-65
        return self._data.copy()

    def tiles(self):
        """Object array of Tile members, the reference's `_data` representation."""
        return np.vectorize(lambda v: _TILE_OF[int(v)], otypes=[object])(self._data)

    def apply(self, converter):                  # map.py:55-58 (host callback over Tile members)
        out = Map(self.width, self.height, 0, 0)
        out._data = np.array([[converter(_TILE_OF[int(self._data[i][j])]) for i in range(self.height + 2)]
                              for j in range(self.width + 2)])
        return out

    def color(self, t, p):                       # map.py:67-81 — one tile through the same GPU encode
        import torch
        from .vec import encode_codes
        buf = torch.full((1, 16), int(t.value), dtype=torch.int8, device="cuda")
        return int(encode_codes(buf, p)[0, 0])

    def state_for_player(self, p):
        """(w+2, h+2) int64 codes: EMPTY 1, WALL -1, own body/slide -2, enemy body/slide -3,
        own head 10, enemy head -10 (map.py:67-84).  Runs on the GPU."""
        import torch
        from .vec import encode_codes
        t = torch.from_numpy(np.ascontiguousarray(self._data)).cuda()
        return encode_codes(t[None], p)[0].cpu().numpy().astype(np.int64)

    def __getitem__(self, index):                # map.py:86-88
        i, j = index
        return _TILE_OF[int(self._data[i + 1][j + 1])]

    def __setitem__(self, position, other):      # map.py:90-92
        i, j = position
        self._data[i + 1][j + 1] = other.value if isinstance(other, Tile) else int(other)
