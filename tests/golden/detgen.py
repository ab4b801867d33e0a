"""Deterministic tensor generation shared by ``make_golden.py`` and the tests.

Weights and inputs of the golden cases are *procedural*: a tensor is a pure
function of (case name, tensor name, shape) through numpy's PCG64 stream, which
is stable across numpy versions and machines.  The fixtures therefore only need
to store names/shapes and the reference's *outputs*, which keeps them small.
"""
import zlib

import numpy as np


def _rng(*parts):
    return np.random.default_rng(zlib.crc32("/".join(str(p) for p in parts).encode()))


def det_normal(case, name, shape, scale=1.0, shift=0.0):
    return (_rng(case, name).standard_normal(tuple(shape)) * scale + shift).astype(np.float32)


def det_param(case, key, shape):
    """Parameter values by role: matrices ~ N(0, 1/fan_in); LayerNorm gains ~ 1 + 0.1 N; biases ~ 0.1 N."""
    shape = tuple(int(s) for s in shape)
    leaf = key.split(".")[-1]
    if key.endswith("pos_embedding"):
        return det_normal(case, key, shape, 0.5)
    if len(shape) >= 2:
        fan_in = int(np.prod(shape[1:]))
        return det_normal(case, key, shape, 1.0 / np.sqrt(fan_in))
    if leaf == "weight":  # LayerNorm gain
        return det_normal(case, key, shape, 0.1, 1.0)
    if leaf == "bg":
        return det_normal(case, key, shape, 0.1, 0.0)
    return det_normal(case, key, shape, 0.1)


def det_state_dict(case, keys, shapes, skip=("inv_freqs",)):
    """{key: np.float32 array} for every key not ending in one of ``skip``."""
    out = {}
    for k, shp in zip(keys, shapes):
        if any(k.endswith(s) for s in skip):
            continue
        out[str(k)] = det_param(case, str(k), shp)
    return out


def sample_index(numel, max_items=384):
    """Deterministic subsample of a flat tensor (all of it when small)."""
    if numel <= max_items:
        return np.arange(numel)
    step = numel // max_items
    return np.arange(0, numel, step)[:max_items]


def sample(arr, max_items=384):
    flat = np.asarray(arr).reshape(-1)
    return flat[sample_index(flat.size, max_items)]


def leading_mask(case, n, L):
    """[n, L] bool rows with 0, 1, L-1 and then pseudo-random counts of leading ones."""
    rng = _rng(case, "mask")
    counts = [0, 1, L - 1] + [int(rng.integers(0, L)) for _ in range(max(0, n - 3))]
    m = np.zeros((n, L), dtype=bool)
    for i, c in enumerate(counts[:n]):
        m[i, :c] = True
    return m


def window_indices(case, n, L, T):
    rng = _rng(case, "idx")
    starts = rng.integers(0, T - L + 1, size=n)
    return (starts[:, None] + np.arange(L)[None, :]).astype(np.int64)
