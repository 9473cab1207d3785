#!/usr/bin/env python
"""One tick of the HEAVIEST chain of BASELINE cfg-4's pool (10M players, 1v1 +-25 + region: the 0-1499 rating group,
3.0M players, 885 passes) alone on one GPU — what the rank that owns it walks under chain sharding.
Run under `rocprofv3 --kernel-trace` to get its passes launch by launch (tools/rocpd_passes.py): the numbers behind the
position-split table.  Usage: python tools/heaviest_chain_tick.py [players] [ticks]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from microservice_matchmaking_amd import Engine, make_config, mode_1v1  # noqa: E402
from microservice_matchmaking_amd.sharding import rating_groups  # noqa: E402
from microservice_matchmaking_amd.synth import make_pool  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
ticks = int(sys.argv[2]) if len(sys.argv) > 2 else 2
cap = 1
while cap < n:
    cap <<= 1
cfg = make_config([mode_1v1(window=25, region_filter=True)], capacity=cap, timing=True)
rating, cons = make_pool(n, seed=1)
sel = rating_groups(cfg, rating) == 0
rating, cons = rating[sel], cons[sel]
with Engine(cfg) as e:
    for k in range(ticks):
        e.reset()
        e.enqueue(rating, cons)
        t0 = time.perf_counter()
        m = e.tick(0)
        dt = time.perf_counter() - t0
        print("tick %d: %d players, %d lobbies, %d passes, walk %.2f ms, tick %.2f ms" %
              (k, len(rating), len(m), m.stats["passes_max"], m.stats["walk_ms"], dt * 1e3), flush=True)
