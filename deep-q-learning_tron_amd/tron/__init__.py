"""tron — MI355X-native TRON light-cycle environment.

Same module names as the reference's `tron/` package (game, map, player, util) so its
trainers import unchanged, plus the batched device API:

    from tron.vec import VecTron, DeviceReplay

All game logic runs in csrc/libtron_hip.so (gfx950 HIP kernels behind the C ABI of
include/tron_hip.h).  There is no CPU implementation in this package.
"""
from . import _native  # noqa: F401

__all__ = ["_native"]
