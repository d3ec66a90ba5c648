"""Text renderer — stands in for the reference's pygame `Window` (tron/window.py:19-37), which
draws each tile with `Tile.color()` (map.py:21-41).  Same `render_map(map)` entry point so
`Game.main_loop(model, pop, window=Window(game))` works on a headless box."""
import sys

from .map import Tile

GLYPH = {
    Tile.EMPTY: ".", Tile.WALL: "#",
    Tile.PLAYER_ONE_BODY: "a", Tile.PLAYER_ONE_HEAD: "A", Tile.PLAYER_ONE_slide: "~",
    Tile.PLAYER_TWO_BODY: "b", Tile.PLAYER_TWO_HEAD: "B", Tile.PLAYER_TWO_slide: "-",
}


def render_ascii(map):
    """One line per storage row of the (w+2) x (h+2) image (border included)."""
    by_value = {t.value: g for t, g in GLYPH.items()}
    return "\n".join("".join(by_value[int(v)] for v in row) for row in map.array())


class Window:
    def __init__(self, game=None, factor=None, stream=None):
        self.game = game
        self.factor = factor            # pixel scale in the reference; unused here
        self.stream = stream or sys.stdout
        self.frames = 0

    def render_map(self, map):
        self.frames += 1
        self.stream.write(render_ascii(map) + "\n\n")
        self.stream.flush()
