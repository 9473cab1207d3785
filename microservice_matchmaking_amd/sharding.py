"""Sharding the search path across the GPUs of one node.

The reference's own partition is the rating group: one queue, one lobby table and one
worker per group, and groups never interact (reference lib/application.ex:26-40,
lib/models/lobby_state.ex:15-29).  So the path shards by rating group with NO data-path
collective: every rank owns one engine and the groups assigned to it, players are routed
by `find_rating_group_by_rating/1` (lib/generic/worker.ex:46-53) exactly as the Generic
worker routes them to per-group AMQP queues.  torch.distributed (RCCL on GPUs, gloo in the
CPU tests) is only used to sum counters / gather results.
"""
from __future__ import annotations

import numpy as np

from ._abi import MMConfig


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


class GroupSharding:
    """group -> rank.  Longest-processing-time-first on `weights` (expected load per group,
    e.g. its share of the pool); ties and the default go round robin."""

    def __init__(self, n_groups: int, world_size: int, weights=None):
        self.n_groups, self.world_size = int(n_groups), int(world_size)
        w = np.ones(n_groups) if weights is None else np.asarray(weights, dtype=np.float64)
        assert w.shape == (n_groups,)
        load = np.zeros(world_size)
        self.owner = np.zeros(n_groups, dtype=np.int64)
        for g in np.argsort(-w, kind="stable"):
            r = int(np.argmin(load))
            self.owner[g] = r
            load[r] += w[g]
        self.load = load

    def groups_of(self, rank: int):
        return [g for g in range(self.n_groups) if self.owner[g] == rank]


class ShardedSearch:
    """One rank's share of the search: its engine + the groups it owns."""

    def __init__(self, cfg: MMConfig, engine_cls, rank: int, world_size: int, weights=None):
        self.cfg, self.rank, self.world_size = cfg, rank, world_size
        self.sharding = GroupSharding(cfg.n_groups, world_size, weights)
        self.engine = engine_cls(cfg)
        self.local_to_global = []          # engine slot -> index in the global arrival order

    def close(self):
        self.engine.close()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def enqueue(self, rating, cons, first_global_index=0):
        """Every rank is handed the same arrival batch (as every Generic worker sees the same
        exchange); it keeps the players whose group it owns, in arrival order."""
        rating = np.asarray(rating, dtype=np.int32)
        cons = np.asarray(cons, dtype=np.uint32)
        grp = rating_groups(self.cfg, rating)
        mine = self.sharding.owner[grp] == self.rank
        idx = np.nonzero(mine)[0]
        slots = self.engine.enqueue(rating[idx], cons[idx], grp[idx])
        for s, i in zip(slots.tolist(), idx.tolist()):
            if s != 0xFFFFFFFF:
                while len(self.local_to_global) <= s:
                    self.local_to_global.append(-1)
                self.local_to_global[s] = first_global_index + i
        return idx, slots

    def tick(self, mode=0):
        return self.engine.tick(mode)

    def global_ids(self, matches):
        """Lobbies of a tick with engine slots translated to global arrival indices."""
        l2g = np.asarray(self.local_to_global, dtype=np.int64)
        return l2g[matches.slots.astype(np.int64)] if len(matches) else np.zeros(matches.slots.shape, np.int64)

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
