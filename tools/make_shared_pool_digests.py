#!/usr/bin/env python
"""tests/golden/shared_pool_digests.json: the CPU oracle's emission digest of the synthetic pools
bench.py runs (BASELINE cfg-2, cfg-3, cfg-4 and their normal-rating variants).

The digest (sharding.union_digest) covers every chain's emission list — the lobbies in publish
order, every lobby its players' global arrival indices in team order — so a bench run on any
number of GPUs can be checked, at full size, against the oracle without the oracle running on
the GPU box.  Test infrastructure: the product never reads the oracle.

    python tools/make_shared_pool_digests.py            # all workloads (about a minute of CPU)
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from bench import DIGESTS, workload_key  # noqa: E402
from microservice_matchmaking_amd.config import make_config, mode_1v1, mode_team  # noqa: E402
from microservice_matchmaking_amd.sharding import tick_digests, union_digest  # noqa: E402
from microservice_matchmaking_amd.synth import ROLE_WEIGHTS_5V5, make_pool  # noqa: E402
from oracle.oracle import OracleEngine, build  # noqa: E402

WORKLOADS = [("1v1", 1_000_000, "uniform"), ("1v1", 1_000_000, "normal"), ("1v1", 10_000_000, "uniform"),
             ("5v5", 1_000_000, "uniform"), ("5v5", 1_000_000, "normal"), ("5v5", 10_000_000, "uniform"),
             ("1v1", 12_000, "uniform"), ("5v5", 12_000, "uniform"), ("1v1", 20_000, "uniform"), ("5v5", 20_000, "uniform")]


def digest_of(mode, n, dist):
    if mode == "1v1":
        modes, kw, window = [mode_1v1(window=25, region_filter=True)], {}, 25
    else:
        modes, kw, window = [mode_team(5, 2, 50, (1, 1, 1, 1, 1))], {"role_weights": ROLE_WEIGHTS_5V5}, 50
    cap = 1
    while cap < n:
        cap <<= 1
    cfg = make_config(modes, capacity=cap, timing=False)
    rating, cons = make_pool(n, seed=1, dist=dist, **kw)
    with OracleEngine(cfg) as eng:
        slots = eng.enqueue(rating, cons)
        assert slots[0] == 0 and slots[-1] == n - 1          # slot == global arrival index
        m = eng.tick(0)
        d = union_digest(tick_digests(0, cfg.n_groups, m.slots.astype(np.int64), m.group))
    return workload_key(mode, n, window, dist), d, int(m.stats["players_matched"])


def main():
    build()
    out = {}
    for mode, n, dist in WORKLOADS:
        key, d, matched = digest_of(mode, n, dist)
        out[key] = d
        print(key, d, "matched", matched, flush=True)
    with open(DIGESTS, "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
        f.write("\n")


if __name__ == "__main__":
    main()
