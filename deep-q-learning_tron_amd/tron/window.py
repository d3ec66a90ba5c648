"""Headless renderers — stand in for the reference's pygame `Window` (tron/window.py:19-37), which
draws each tile with `Tile.color()` (map.py:21-41).  Same `render_map(map)` entry point so
`Game.main_loop(model, pop, window=Window(game))` works on a box without a display: frames go out
as text, and — with `png_dir` — as PNG images with the reference's colours and geometry."""
import os
import struct
import sys
import zlib

import numpy as np

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


def render_rgb(map, factor=10):
    """The frame window.py:19-37 draws, as a uint8 [(h+2)*factor, (w+2)*factor, 3] image: white
    screen, black board at (factor, factor), then every interior tile as a factor x factor square in
    Tile.color() at pixel ((col + 1.1) * factor, (row + 1.1) * factor) — the reference's 0.1 offset
    included (pygame truncates rect coordinates to integers)."""
    w, h = map.width, map.height
    img = np.full(((h + 2) * factor, (w + 2) * factor, 3), 255, np.uint8)
    img[factor:factor + h * factor, factor:factor + w * factor] = 0
    by_value = {t.value: t for t in Tile}
    for row in range(h):
        for col in range(w):
            x, y = int((col + 1.1) * factor), int((row + 1.1) * factor)
            img[y:y + factor, x:x + factor] = by_value[int(map.array()[row + 1, col + 1])].color()
    return img


def write_png(path, rgb):
    """Minimal PNG encoder (8-bit RGB, one IDAT) — no imaging library in the image."""
    h, w, _ = rgb.shape
    raw = b"".join(b"\x00" + rgb[y].tobytes() for y in range(h))

    def chunk(kind, data):
        return struct.pack(">I", len(data)) + kind + data + struct.pack(">I", zlib.crc32(kind + data) & 0xFFFFFFFF)
    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 2, 0, 0, 0)) +
                chunk(b"IDAT", zlib.compress(raw, 6)) + chunk(b"IEND", b""))


class Window:
    def __init__(self, game=None, factor=None, stream=None, png_dir=None):
        self.game = game
        self.factor = factor or 10      # pixel scale (window.py:21)
        self.stream = stream or sys.stdout
        self.png_dir = png_dir
        self.frames = 0
        if png_dir:
            os.makedirs(png_dir, exist_ok=True)

    def render_map(self, map):
        self.frames += 1
        self.stream.write(render_ascii(map) + "\n\n")
        self.stream.flush()
        if self.png_dir:
            write_png(os.path.join(self.png_dir, "frame_%05d.png" % self.frames), render_rgb(map, self.factor))
