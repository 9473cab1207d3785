"""Seeded synthetic player pools.

Counter-based SplitMix64 in pure integer arithmetic, so the same (seed, index) gives the
same player in numpy here and in any C restatement.  The clipped-normal rating uses an
Irwin–Hall sum of twelve 32-bit uniforms (no libm → bit-exact everywhere).
"""
from __future__ import annotations

import numpy as np

from ._abi import cons_make

_GOLD = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)


def splitmix64(seed, idx):
    """value(seed, idx) = mix(seed + (idx + 1) * golden); idx may be an array."""
    with np.errstate(over="ignore"):
        z = np.uint64(seed) + (np.asarray(idx, dtype=np.uint64) + np.uint64(1)) * _GOLD
        z = (z ^ (z >> np.uint64(30))) * _M1
        z = (z ^ (z >> np.uint64(27))) * _M2
        return z ^ (z >> np.uint64(31))


def _stream(seed, stream, n):
    # independent streams: mix the stream id into the seed first
    s = int(splitmix64(seed, np.uint64(0xC0FFEE) + np.uint64(stream)))
    return splitmix64(s, np.arange(n, dtype=np.uint64))


def ratings_uniform(n, seed, lo=0, hi=5000):
    return (lo + (_stream(seed, 1, n) % np.uint64(hi - lo + 1)).astype(np.int64)).astype(np.int32)


def ratings_normal(n, seed, mu=2500, sigma=700, lo=0, hi=5000):
    """mu + sigma * (sum of 12 U32 / 2^32 - 6), floored, clipped to [lo, hi]."""
    acc = np.zeros(n, dtype=np.int64)
    for k in range(12):
        acc += (_stream(seed, 100 + k, n) >> np.uint64(32)).astype(np.int64)
    z = acc - (np.int64(6) << np.int64(32))
    r = mu + ((sigma * z) >> np.int64(32))          # arithmetic shift = floor
    return np.clip(r, lo, hi).astype(np.int32)


def choice_weighted(n, seed, stream, weights):
    """Index drawn with integer weights (e.g. roles {15,15,30,30,10})."""
    w = np.asarray(weights, dtype=np.uint64)
    u = _stream(seed, stream, n) % np.uint64(int(w.sum()))
    edges = np.cumsum(w)
    return np.searchsorted(edges, u, side="right").astype(np.uint32)


def make_pool(n, seed=1, dist="uniform", n_regions=8, mode=0, role_weights=None,
              mode_weights=None, party_max=1):
    """Returns (rating int32[n], cons uint32[n]); arrival order = index."""
    rating = ratings_uniform(n, seed) if dist == "uniform" else ratings_normal(n, seed)
    region = (_stream(seed, 2, n) % np.uint64(max(1, n_regions))).astype(np.uint32)
    role = choice_weighted(n, seed, 3, role_weights) if role_weights else np.zeros(n, np.uint32)
    if mode_weights:
        md = choice_weighted(n, seed, 4, mode_weights)
    else:
        md = np.full(n, mode, np.uint32)
    party = (1 + _stream(seed, 5, n) % np.uint64(party_max)).astype(np.uint32) if party_max > 1 \
        else np.zeros(n, np.uint32)
    return rating, cons_make(md, region, party, role)


ROLE_WEIGHTS_5V5 = (15, 15, 30, 30, 10)
