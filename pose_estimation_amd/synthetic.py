"""Seeded synthetic inputs of the benchmark shapes (NumPy PCG64; no datasets needed).

The reference's own inputs are private TFRecords (test.ipynb:59); these stand in for them.
"""
from __future__ import annotations

import numpy as np


def decode_heatmaps(B: int, K: int, H: int, W: int, seed: int = 0) -> np.ndarray:
    """float32 [B,K,H,W]: one Gaussian blob (sigma 2, amplitude U(0.3,1)) per map at a
    uniformly random sub-pixel centre, plus N(0, 0.01) noise."""
    rng = np.random.Generator(np.random.PCG64(seed))
    cx = rng.uniform(0, W, size=(B, K, 1, 1)).astype(np.float32)
    cy = rng.uniform(0, H, size=(B, K, 1, 1)).astype(np.float32)
    amp = rng.uniform(0.3, 1.0, size=(B, K, 1, 1)).astype(np.float32)
    xs = np.arange(W, dtype=np.float32)[None, None, None, :]
    ys = np.arange(H, dtype=np.float32)[None, None, :, None]
    hm = amp * np.exp(-((xs - cx) ** 2 + (ys - cy) ** 2) / np.float32(8.0))
    hm = hm + rng.normal(0.0, 0.01, size=hm.shape).astype(np.float32)
    return hm.astype(np.float32)


def joints(B: int, K: int, in_h: int, in_w: int, seed: int = 2) -> np.ndarray:
    """float32 [B,K,3]: x ~ U(0,in_w), y ~ U(0,in_h), v in {0,1,2} w.p. (0.1,0.2,0.7)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    x = rng.uniform(0, in_w, size=(B, K))
    y = rng.uniform(0, in_h, size=(B, K))
    v = rng.choice(3, size=(B, K), p=[0.1, 0.2, 0.7])
    return np.stack([x, y, v], axis=-1).astype(np.float32)


def images_u8(B: int, in_h: int, in_w: int, seed: int = 1) -> np.ndarray:
    """uint8 [B,in_h,in_w,3] uniform noise."""
    rng = np.random.Generator(np.random.PCG64(seed))
    return rng.integers(0, 256, size=(B, in_h, in_w, 3), dtype=np.uint8)


def ignore_mask(B: int, H: int, W: int, seed: int = 4, p_ignore: float = 0.1) -> np.ndarray:
    """uint8 [B,H,W]: 0 = ignore with probability p_ignore, else 1."""
    rng = np.random.Generator(np.random.PCG64(seed))
    return (rng.uniform(size=(B, H, W)) >= p_ignore).astype(np.uint8)
