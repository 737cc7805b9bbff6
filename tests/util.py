"""Shared helpers for the test-suite (oracle access, golden fixtures, random graphs)."""
from __future__ import annotations

import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")

GCN_WIDTHS = (1, 4, 7, 16, 32, 64, 100, 128, 256, 300)
GAT_SHAPES = ((1, 7), (2, 4), (8, 8), (8, 64))


def golden(name: str):
    return np.load(os.path.join(GOLDEN, name))


def random_graph(seed: int, n: int, e: int, duplicates: bool = False, hub: bool = True):
    """Seeded directed multigraph-free (unless ``duplicates``) edge arrays in random order."""
    rng = np.random.default_rng(seed)
    if duplicates:
        src = rng.integers(0, n, e, dtype=np.int64)
        dst = rng.integers(0, n, e, dtype=np.int64)
    else:
        total = n * n
        if e > total:
            raise ValueError("too many edges")
        if total <= 4 * e or total < (1 << 22):
            keys = rng.choice(total, size=e, replace=False)
        else:
            keys = np.unique(rng.integers(0, total, int(e * 1.2) + 16, dtype=np.int64))
            rng.shuffle(keys)
            keys = keys[:e]
            assert keys.shape[0] == e
        src, dst = keys // n, keys % n
    if hub and n > 4 and e > 8:           # a high-degree destination to exercise the ragged path
        k = max(1, e // 8)
        dst[:k] = 1
        if not duplicates:
            pair = np.unique(np.stack([src, dst], 1), axis=0)
            rng.shuffle(pair)
            src, dst = pair[:, 0], pair[:, 1]
    return src.astype(np.int32), dst.astype(np.int32)


def gcn_norm(in_degrees: np.ndarray) -> np.ndarray:
    """norm = in_deg^-0.5, inf -> 0  (benchmarking/gcn/seastar/train.py:53-57)."""
    with np.errstate(divide="ignore"):
        norm = np.power(in_degrees.astype(np.float32), np.float32(-0.5))
    norm[np.isinf(norm)] = 0
    return norm.astype(np.float32).reshape(-1, 1)
