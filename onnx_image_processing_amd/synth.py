"""Deterministic synthetic gray image pairs.

The generator is a pure integer hash (splitmix64 finaliser), so the same
pixels come out of numpy here, on the GPU box, and in any later re-run; nothing
is read from disk.  Images are uint8-valued: the reference's callers feed
uint8 camera frames converted to float32 in [0, 255]
(reference sample/image_matching.py:42-46).

image(seed)  = coarse 8x8 blocks with values 0..199  +  per-pixel texture 0..54
pair(seed)   = (image, image circularly shifted by (dy, dx) [+ optional +-noise])
"""
from __future__ import annotations

import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _mix64(z: np.ndarray) -> np.ndarray:
    """splitmix64 output function on a uint64 array (wrap-around arithmetic)."""
    z = z.astype(np.uint64, copy=True)
    with np.errstate(over="ignore"):
        z += np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return z


def _hash3(seed: int, a: np.ndarray, b: np.ndarray, salt: int) -> np.ndarray:
    with np.errstate(over="ignore"):
        k = (
            np.uint64(seed) * np.uint64(0xD1342543DE82EF95)
            + a.astype(np.uint64) * np.uint64(0x2545F4914F6CDD1D)
            + b.astype(np.uint64) * np.uint64(0x9FB21C651E98DF25)
            + np.uint64(salt)
        )
    return _mix64(_mix64(k))


def synth_image(seed: int, height: int = 480, width: int = 640) -> np.ndarray:
    """uint8 (H, W) image for `seed`."""
    y = np.arange(height, dtype=np.uint64)[:, None]
    x = np.arange(width, dtype=np.uint64)[None, :]
    yy, xx = np.broadcast_arrays(y, x)
    coarse = _hash3(seed, yy // np.uint64(8), xx // np.uint64(8), 1) % np.uint64(200)
    fine = _hash3(seed, yy, xx, 2) % np.uint64(55)
    return np.minimum(coarse + fine, np.uint64(255)).astype(np.uint8)


def synth_pair(
    seed: int,
    height: int = 480,
    width: int = 640,
    shift: tuple[int, int] = (3, 5),
    noise: int = 0,
) -> tuple[np.ndarray, np.ndarray]:
    """uint8 image pair: second image = first rolled by `shift` (+ optional noise)."""
    img1 = synth_image(seed, height, width)
    img2 = np.roll(img1, shift=shift, axis=(0, 1))
    if noise > 0:
        y = np.arange(height, dtype=np.uint64)[:, None]
        x = np.arange(width, dtype=np.uint64)[None, :]
        yy, xx = np.broadcast_arrays(y, x)
        n = (_hash3(seed + 1, yy, xx, 3) % np.uint64(2 * noise + 1)).astype(np.int64) - noise
        img2 = np.clip(img2.astype(np.int64) + n, 0, 255).astype(np.uint8)
    return img1, img2


def synth_batch(
    first_seed: int,
    num_pairs: int,
    height: int = 480,
    width: int = 640,
    shift: tuple[int, int] = (3, 5),
    noise: int = 0,
) -> tuple[np.ndarray, np.ndarray]:
    """float32 (B,1,H,W) x2 in [0,255]; pair i uses seed first_seed + i."""
    a = np.empty((num_pairs, 1, height, width), np.float32)
    b = np.empty((num_pairs, 1, height, width), np.float32)
    for i in range(num_pairs):
        p, q = synth_pair(first_seed + i, height, width, shift, noise)
        a[i, 0] = p
        b[i, 0] = q
    return a, b


def synth_batch_u8(
    first_seed: int,
    num_pairs: int,
    height: int = 480,
    width: int = 640,
    shift: tuple[int, int] = (3, 5),
    noise: int = 0,
) -> tuple[np.ndarray, np.ndarray]:
    """uint8 (B,1,H,W) x2: the same frames as synth_batch before the float32 conversion (what a camera delivers;
    the u8 ingest path takes them as they are)."""
    a = np.empty((num_pairs, 1, height, width), np.uint8)
    b = np.empty((num_pairs, 1, height, width), np.uint8)
    for i in range(num_pairs):
        a[i, 0], b[i, 0] = synth_pair(first_seed + i, height, width, shift, noise)
    return a, b
