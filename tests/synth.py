"""Seeded synthetic images: counter-based hash, no fixtures needed.

byte(i, y, x, c) = splitmix64(seed ^ (i << 40 | y << 24 | x << 3 | c)) & 0xFF
"""
import numpy as np

_M = np.uint64(0xFFFFFFFFFFFFFFFF)


def splitmix64(z):
    z = (z + np.uint64(0x9E3779B97F4A7C15)) & _M
    z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M
    z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M
    return z ^ (z >> np.uint64(31))


def uniform(h, w, c, seed=0xFA171200, index=0):
    """Distribution U: i.i.d. uniform bytes."""
    with np.errstate(over="ignore"):
        y = np.arange(h, dtype=np.uint64)[:, None, None] << np.uint64(24)
        x = np.arange(w, dtype=np.uint64)[None, :, None] << np.uint64(3)
        ch = np.arange(c, dtype=np.uint64)[None, None, :]
        key = (np.uint64(index) << np.uint64(40)) | y | x | ch
        return (splitmix64(np.uint64(seed) ^ key) & np.uint64(0xFF)).astype(np.uint8)


def photo(h, w, c, seed=0xFA171200, index=0):
    """Distribution P: low-frequency cosine mix plus +-8 LSB of hash noise."""
    yy = np.arange(h, dtype=np.float64)[:, None, None]
    xx = np.arange(w, dtype=np.float64)[None, :, None]
    cc = np.arange(c, dtype=np.float64)[None, None, :]
    base = 127.5 + 60 * np.cos(2 * np.pi * (xx / max(w, 1) * (1.5 + cc) + index * 0.1)) \
                 + 50 * np.cos(2 * np.pi * (yy / max(h, 1) * (2.5 - 0.5 * cc))) \
                 + 15 * np.cos(2 * np.pi * ((xx + 2 * yy) / 37.0))
    noise = (uniform(h, w, c, seed ^ 0x5555, index).astype(np.float64) / 255.0 - 0.5) * 16.0
    return np.clip(np.rint(base + noise), 0, 255).astype(np.uint8)


def edges(h, w, c):
    """Distribution E: all-0, all-255, 1-px checkerboard, single white pixel."""
    zero = np.zeros((h, w, c), np.uint8)
    full = np.full((h, w, c), 255, np.uint8)
    yy, xx = np.mgrid[0:h, 0:w]
    checker = (((yy + xx) & 1) * 255).astype(np.uint8)[:, :, None].repeat(c, axis=2)
    impulse = zero.copy()
    impulse[h // 2, w // 2, :] = 255
    return {"zero": zero, "full": full, "checker": checker, "impulse": impulse}
