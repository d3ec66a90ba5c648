"""Deterministic network weights shared by the fixture generator and the tests: an integer
hash per element, so no RNG-library or libm differences between machines."""
import numpy as np


def det_tensor(shape, key_index, scale):
    n = int(np.prod(shape))
    idx = np.arange(n, dtype=np.uint64)
    with np.errstate(over="ignore"):
        x = idx * np.uint64(6364136223846793005) + np.uint64(1442695040888963407) \
            + np.uint64(key_index) * np.uint64(0x9E3779B97F4A7C15)
        x ^= x >> np.uint64(29)
        x = x * np.uint64(0xBF58476D1CE4E5B9)
        x ^= x >> np.uint64(32)
    v = ((x >> np.uint64(20)) % np.uint64(2001)).astype(np.int64) - 1000
    return (v.astype(np.float64) / 1000.0 * scale).astype(np.float32).reshape(shape)


def det_state_dict(shapes, salt=0):
    """shapes: ordered {name: shape}.  weights ~ U(-1,1)/sqrt(fan_in), biases ~ U(-0.1,0.1)."""
    import torch
    out = {}
    for k, (name, shape) in enumerate(shapes.items()):
        shape = tuple(shape)
        if name.endswith("bias"):
            scale = 0.1
        else:
            fan_in = int(np.prod(shape[1:]))
            scale = 1.0 / np.sqrt(fan_in)
        out[name] = torch.from_numpy(det_tensor(shape, k + 100 * salt, scale))
    return out


def mm_stream(seed, n=96):
    """The u32 draw stream of one minimax fixture case: fmix32(seed + i * golden) (murmur3 finaliser)."""
    x = (np.uint64(seed) + np.arange(n, dtype=np.uint64) * np.uint64(0x9E3779B9)) & np.uint64(0xFFFFFFFF)
    x ^= x >> np.uint64(16)
    x = (x * np.uint64(0x85EBCA6B)) & np.uint64(0xFFFFFFFF)
    x ^= x >> np.uint64(13)
    x = (x * np.uint64(0xC2B2AE35)) & np.uint64(0xFFFFFFFF)
    x ^= x >> np.uint64(16)
    return x.astype(np.uint32)
