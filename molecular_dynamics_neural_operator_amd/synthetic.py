"""Seeded synthetic MD frames in the reference's frame convention.

One frame = ``float32 [n_atoms, 3]`` in Angstrom, frames in trajectory order
(dataset.py:159 after the ``[T,3,N] -> [T,N,3]`` transpose; preprocess.py:51 for the
all-atom selection).  There is no network and no BBA data in the image, so every test,
fixture and benchmark input comes from here (SURVEY.md §8d "synthetic inputs").

Shapes
  A  BBA C-alpha as in the reference: N=28 random-walk chain, 3.8 A steps, r=8 A
  B  BBA all-atom stand-in: N=504 uniform in a cube at 0.1 atoms/A^3 (L=17.1 A), r=8 A
  C  50k-atom box: N=50,000 uniform, L=79.4 A, r=10 A
"""
from __future__ import annotations

import numpy as np

NUM_AMINO_ACIDS = 20  # Embedding(20, 4), graph_kernel.py:267


def chain_frame(n_atoms: int = 28, step: float = 3.8, seed: int = 0) -> np.ndarray:
    """Random-walk C-alpha chain: consecutive atoms `step` Angstrom apart."""
    rng = np.random.default_rng(seed)
    d = rng.normal(size=(n_atoms, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    pos = np.cumsum(d * step, axis=0)
    pos -= pos.mean(axis=0, keepdims=True)
    return pos.astype(np.float32)


def box_frame(n_atoms: int = 504, density: float = 0.1, seed: int = 1) -> np.ndarray:
    """Uniform atoms in a non-periodic cube of side (n/density)^(1/3), origin-centred."""
    rng = np.random.default_rng(seed)
    side = (n_atoms / density) ** (1.0 / 3.0)
    return ((rng.random((n_atoms, 3)) - 0.5) * side).astype(np.float32)


def jitter_window(base: np.ndarray, window: int = 10, sigma: float = 0.05, seed: int = 0) -> np.ndarray:
    """`window` frames = base + N(0, sigma^2) per frame -> float32 [W, N, 3]."""
    rng = np.random.default_rng(seed + 7919)
    noise = rng.normal(scale=sigma, size=(window,) + base.shape)
    return (base[None] + noise).astype(np.float32)


def ou_trajectory(base: np.ndarray, n_frames: int, sigma: float = 0.3, theta: float = 0.1,
                  seed: int = 2) -> np.ndarray:
    """Ornstein-Uhlenbeck jitter around `base` -> float32 [T, N, 3] (cfg4 training data)."""
    rng = np.random.default_rng(seed)
    out = np.empty((n_frames,) + base.shape, dtype=np.float32)
    dev = np.zeros_like(base, dtype=np.float64)
    for t in range(n_frames):
        dev += -theta * dev + sigma * np.sqrt(2 * theta) * rng.normal(size=base.shape)
        out[t] = (base + dev).astype(np.float32)
    return out


def amino_acids(n_atoms: int, seed: int = 0) -> np.ndarray:
    """int64 [N] residue-type ids in [0, 20)."""
    rng = np.random.default_rng(seed + 104729)
    return rng.integers(0, NUM_AMINO_ACIDS, size=n_atoms, dtype=np.int64)


def ensemble_windows(base_window: np.ndarray, members: int, sigma: float = 0.1,
                     seed0: int = 100) -> np.ndarray:
    """cfg3: member m = base window + N(0, sigma^2) with seed seed0+m -> [M, W, N, 3]."""
    out = np.empty((members,) + base_window.shape, dtype=np.float32)
    for m in range(members):
        rng = np.random.default_rng(seed0 + m)
        out[m] = base_window + rng.normal(scale=sigma, size=base_window.shape).astype(np.float32)
    return out


def min_threshold_gap(frame: np.ndarray, threshold: float) -> float:
    """Smallest | ||p_i-p_j|| - threshold | over all pairs (f64), for near-threshold screening."""
    p = frame.astype(np.float64)
    d = np.sqrt(((p[:, None, :] - p[None, :, :]) ** 2).sum(-1))
    return float(np.abs(d - threshold).min())


def contact_map(frame: np.ndarray, cutoff: float = 8.0) -> np.ndarray:
    """Flat [rows..., cols...] contact map of one frame as the reference's data files store it
    (dataset.py:114, 189): pairs with f64 distance < cutoff, self-pairs included, row-major order."""
    d = frame.astype(np.float64)
    dist = np.sqrt(((d[:, None, :] - d[None, :, :]) ** 2).sum(-1))
    rows, cols = np.nonzero(dist < cutoff)
    return np.concatenate([rows, cols]).astype(np.int64)

