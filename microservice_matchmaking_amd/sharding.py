"""Sharding the search path across the GPUs of one node.

The reference's own partition is the rating group: one queue, one lobby table and one
worker per group, and groups never interact (reference lib/application.ex:26-40,
lib/models/lobby_state.ex:15-29).  Inside a group the stored lobby is selected by
`game_mode` (lib/models/lobby_state.ex:74-83), so two modes of one group never share a lobby
either: the unit that cannot be split is the CHAIN = (game mode, rating group), and the shard
key is exactly that pair.  The path therefore shards with NO data-path collective: every
rank owns one engine and the chains assigned to it, players are routed by
`find_rating_group_by_rating/1` (lib/generic/worker.ex:46-53) and their `"game-mode"`
(lib/search/worker.ex:294) exactly as the Generic worker routes them to per-group AMQP
queues.  torch.distributed (RCCL on GPUs, gloo in the CPU tests) is only used to sum
counters / gather results.

What this does NOT do, and why: split ONE chain across ranks with a
rating-bucket halo all-gather.  A chain has one open lobby (lobby_state.ex:90-91) and one
cursor; every pass of the cursor depends on the pass before, so a split chain would hop
between devices once per pass (hundreds of passes a tick) — there is no order-free boundary
region to exchange.  A pool of G groups and M modes therefore uses at most G x M ranks, and
the tick of the pool takes as long as its heaviest chain.
"""
from __future__ import annotations

import hashlib

import numpy as np

from ._abi import MMConfig, NO_SLOT


def rating_groups(cfg: MMConfig, rating) -> np.ndarray:
    """Vectorised find_rating_group_by_rating/1: first inclusive range in table order, else
    the default group (also for NaN = a non-number JSON value)."""
    r = np.asarray(rating, dtype=np.float64)
    out = np.full(r.shape, int(cfg.default_group), dtype=np.uint8)
    done = np.isnan(r)
    for g in range(cfg.n_groups):
        hit = (~done) & (r >= cfg.groups[g].from_) & (r <= cfg.groups[g].to)
        out[hit] = g
        done |= hit
    return out


def chain_weights(cfg: MMConfig, rating, cons) -> np.ndarray:
    """Players per (mode, group) chain of a pool: the load estimate ChainSharding balances."""
    grp = rating_groups(cfg, rating).astype(np.int64)
    mode = (np.asarray(cons, dtype=np.uint32) & 0xF).astype(np.int64)
    ok = mode < cfg.n_modes
    flat = np.bincount(mode[ok] * cfg.n_groups + grp[ok], minlength=cfg.n_modes * cfg.n_groups)
    return flat.reshape(cfg.n_modes, cfg.n_groups).astype(np.float64)


class ChainSharding:
    """(mode, group) -> rank.  Longest-processing-time-first on `weights` (expected load per
    chain, e.g. its share of the pool); equal weights go round robin.  With fewer chains than
    ranks the surplus ranks stay idle (`idle_ranks`) — 7 groups of one mode on 8 GPUs leave
    one GPU without work, by construction of the reference's partition."""

    def __init__(self, n_modes: int, n_groups: int, world_size: int, weights=None):
        self.n_modes, self.n_groups, self.world_size = int(n_modes), int(n_groups), int(world_size)
        if weights is None:
            w = np.ones((self.n_modes, self.n_groups))
        else:
            w = np.asarray(weights, dtype=np.float64)
            if w.ndim == 1:                       # one row of group weights for every mode
                w = np.broadcast_to(w, (self.n_modes, self.n_groups)).copy()
        assert w.shape == (self.n_modes, self.n_groups), w.shape
        self.weights = w
        load = np.zeros(self.world_size)
        owner = np.zeros(self.n_modes * self.n_groups, dtype=np.int64)
        for c in np.argsort(-w.ravel(), kind="stable"):
            r = int(np.argmin(load))
            owner[c] = r
            load[r] += max(w.ravel()[c], 1e-12)   # weightless chains still spread out
        self.chain_owner = owner.reshape(self.n_modes, self.n_groups)
        self.load = np.array([w[self.chain_owner == r].sum() for r in range(self.world_size)])

    def chains_of(self, rank: int):
        return [(m, g) for m in range(self.n_modes) for g in range(self.n_groups)
                if self.chain_owner[m, g] == rank]

    def idle_ranks(self):
        used = set(int(r) for r in self.chain_owner.ravel())
        return [r for r in range(self.world_size) if r not in used]

    def bound(self):
        """Total load / heaviest rank.  NOT the speed-up to expect: one GPU already runs all chains concurrently,
        so the pool's step on one GPU is close to its heaviest chain's and sharding gains far less
        (bench.predict_speedup)."""
        top = float(self.load.max())
        return float(self.weights.sum() / top) if top > 0 else 1.0

    def describe(self):
        return {"key": "(game mode, rating group)", "world_size": self.world_size,
                "owner": self.chain_owner.tolist(), "load_share": (self.load / max(self.load.sum(), 1e-12)).tolist(),
                "idle_ranks": self.idle_ranks(), "load_share_bound": self.bound()}


class GroupSharding(ChainSharding):
    """One mode: group -> rank (the round-1 interface, kept for its callers)."""

    def __init__(self, n_groups: int, world_size: int, weights=None):
        super().__init__(1, n_groups, world_size, weights)
        self.owner = self.chain_owner[0]

    def groups_of(self, rank: int):
        return [g for g in range(self.n_groups) if self.owner[g] == rank]


def chain_digest(ids: np.ndarray) -> str:
    """Digest of ONE chain's emission list: the lobbies in publish order, every lobby its
    players' global arrival indices in team order (int64 little endian)."""
    return hashlib.blake2b(np.ascontiguousarray(ids, dtype="<i8").tobytes(), digest_size=16).hexdigest()


def union_digest(per_chain: dict) -> str:
    """Digest of the whole pool's emission: chain digests in (mode, group) order.  A chain that
    emitted nothing contributes the digest of the empty list, so the value does not depend on
    which rank (if any) owned it."""
    h = hashlib.blake2b(digest_size=16)
    for key in sorted(per_chain):
        h.update(("%d/%d:%s;" % (key[0], key[1], per_chain[key])).encode())
    return h.hexdigest()


def tick_digests(mode: int, n_groups: int, ids: np.ndarray, group: np.ndarray) -> dict:
    """{(mode, g): chain_digest} for every group of one tick's match list (ids = global indices)."""
    out = {}
    group = np.asarray(group)
    for g in range(n_groups):
        sel = ids[group == g] if len(ids) else np.zeros((0,), np.int64)
        out[(mode, g)] = chain_digest(sel)
    return out


class ShardedSearch:
    """One rank's share of the search: its engine + the chains it owns."""

    def __init__(self, cfg: MMConfig, engine_cls, rank: int, world_size: int, weights=None):
        self.cfg, self.rank, self.world_size = cfg, rank, world_size
        self.sharding = ChainSharding(cfg.n_modes, cfg.n_groups, world_size, weights)
        self.engine = engine_cls(cfg)
        self.local_to_global = np.full(int(cfg.capacity), -1, dtype=np.int64)   # engine slot -> global arrival index

    def close(self):
        self.engine.close()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def mine(self, rating, cons):
        """Mask of the players of an arrival batch whose chain this rank owns."""
        grp = rating_groups(self.cfg, rating)
        mode = (np.asarray(cons, dtype=np.uint32) & 0xF).astype(np.int64)
        ok = mode < self.cfg.n_modes                      # unknown mode: nobody's (a single engine rejects it)
        own = np.zeros(len(grp), dtype=bool)
        own[ok] = self.sharding.chain_owner[mode[ok], grp[ok].astype(np.int64)] == self.rank
        return own, grp

    def enqueue(self, rating, cons, first_global_index=0):
        """Every rank is handed the same arrival batch (as every Generic worker sees the same
        exchange); it keeps the players whose chain it owns, in arrival order."""
        rating = np.asarray(rating, dtype=np.int32)
        cons = np.asarray(cons, dtype=np.uint32)
        own, grp = self.mine(rating, cons)
        idx = np.nonzero(own)[0]
        slots = self.engine.enqueue(rating[idx], cons[idx], grp[idx])
        ok = slots != NO_SLOT
        self.local_to_global[slots[ok].astype(np.int64)] = first_global_index + idx[ok]
        return idx, slots

    def tick(self, mode=0):
        return self.engine.tick(mode)

    def global_ids(self, matches):
        """Lobbies of a tick with engine slots translated to global arrival indices."""
        if not len(matches):
            return np.zeros(matches.slots.shape, np.int64)
        return self.local_to_global[matches.slots.astype(np.int64)]

    @staticmethod
    def sum_over_ranks(values):
        """Counters summed over the ranks (the only collective of the path)."""
        import torch
        import torch.distributed as dist
        t = torch.tensor([float(v) for v in values], dtype=torch.float64)
        if dist.is_available() and dist.is_initialized():
            if dist.get_backend() == "nccl":
                t = t.cuda()
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return t.cpu().tolist()
