#!/usr/bin/env python
"""tests/golden/shared_pool_digests.json: the CPU oracle's emission digest of the synthetic pools
bench.py runs (BASELINE cfg-2, cfg-3, cfg-4 and their normal-rating variants).

The digest (sharding.union_digest) covers every chain's emission list — the lobbies in publish
order, every lobby its players' global arrival indices in team order — so a bench run on any
number of GPUs can be checked, at full size, against the oracle without the oracle running on
the GPU box.  Test infrastructure: the product never reads the oracle.

The stream legs of bench.py (BASELINE cfg-5 and its 1v1-only variant) are tick-count driven and therefore
deterministic (microservice_matchmaking_amd/stream.py): their keys hold the oracle's union digest of everything
the stream emitted, the matched players and the backlog per mode and group at the end, for the default 3 s legs
and for cfg-5's stated 60 s.

    python tools/make_shared_pool_digests.py            # all workloads (about three minutes of CPU)
    python tools/make_shared_pool_digests.py stream     # only the stream keys
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from bench import DIGESTS, stream_capacity, stream_key, workload_key  # noqa: E402
from microservice_matchmaking_amd.config import make_config, mode_1v1, mode_team  # noqa: E402
from microservice_matchmaking_amd.sharding import ShardedSearch, tick_digests, union_digest  # noqa: E402
from microservice_matchmaking_amd.stream import latency_summary, run_stream, stream_schedule  # noqa: E402
from microservice_matchmaking_amd.synth import ROLE_WEIGHTS_5V5, make_pool  # noqa: E402
from oracle.oracle import OracleEngine, build  # noqa: E402

WORKLOADS = [("1v1", 1_000_000, "uniform"), ("1v1", 1_000_000, "normal"), ("1v1", 10_000_000, "uniform"),
             ("5v5", 1_000_000, "uniform"), ("5v5", 1_000_000, "normal"), ("5v5", 10_000_000, "uniform"),
             ("1v1", 12_000, "uniform"), ("5v5", 12_000, "uniform"), ("1v1", 20_000, "uniform"), ("5v5", 20_000, "uniform"),
             # bench.py's concurrent_pools leg: pool k of the leg is seeded 101 + k (up to four pools)
             ("1v1", 1_000_000, "uniform", 101), ("1v1", 1_000_000, "uniform", 102), ("1v1", 1_000_000, "uniform", 103),
             ("1v1", 1_000_000, "uniform", 104), ("5v5", 1_000_000, "uniform", 101), ("5v5", 1_000_000, "uniform", 102),
             ("1v1", 12_000, "uniform", 101), ("1v1", 12_000, "uniform", 102)]


def digest_of(mode, n, dist, seed=1):
    if mode == "1v1":
        modes, kw, window = [mode_1v1(window=25, region_filter=True)], {}, 25
    else:
        modes, kw, window = [mode_team(5, 2, 50, (1, 1, 1, 1, 1))], {"role_weights": ROLE_WEIGHTS_5V5}, 50
    cap = 1
    while cap < n:
        cap <<= 1
    cfg = make_config(modes, capacity=cap, timing=False)
    rating, cons = make_pool(n, seed=seed, dist=dist, **kw)
    with OracleEngine(cfg) as eng:
        slots = eng.enqueue(rating, cons)
        assert slots[0] == 0 and slots[-1] == n - 1          # slot == global arrival index
        m = eng.tick(0)
        d = union_digest(tick_digests(0, cfg.n_groups, m.slots.astype(np.int64), m.group))
    return workload_key(mode, n, window, dist, seed), d, int(m.stats["players_matched"])


# (label, seconds, players/s), 10 ms ticks: bench.py's default legs, cfg-5 at its stated size, the dry runs of tests/
STREAMS = [("1v1", 3.0, 100_000), ("mixed", 3.0, 100_000), ("1v1", 60.0, 100_000), ("mixed", 60.0, 100_000),
           ("1v1", 0.1, 20_000), ("mixed", 0.1, 20_000),
           # bench.py's latency_saturation leg (1 s per rate): the rate whose run is checked against the oracle's
           ("1v1", 1.0, 1_000_000), ("1v1", 1.0, 100_000)]


def stream_digest_of(label, seconds, qps=100_000, tick_ms=10.0, seed=77):
    """The oracle driven through exactly the schedule bench.py's stream legs use."""
    w25 = mode_1v1(window=25, region_filter=True)
    cap = stream_capacity(qps, seconds)
    if label == "mixed":
        cfg = make_config([w25, mode_team(5, 2, 50, (1, 1, 1, 1, 1))], capacity=cap, timing=False)
        kw = {"mode_weights": (70, 30), "role_weights": ROLE_WEIGHTS_5V5}
    else:
        cfg = make_config([w25], capacity=cap, timing=False)
        kw = {}
    with ShardedSearch(cfg, OracleEngine, 0, 1) as s:
        res = run_stream(s, stream_schedule(qps, seconds, tick_ms, seed), realtime=False, **kw)
    floors = [latency_summary(res["floor"][md], res["floor"][md]) for md in range(int(cfg.n_modes))]
    return stream_key(label, qps, seconds, tick_ms, seed), {
        "digest": union_digest(res["digests"]), "matched": int(res["matched"]), "arrivals": int(res["arrivals"]),
        "backlog": [d.tolist() for d in res["depth"]],
        "floor_p50_ms": [f["p50_ms"] for f in floors], "floor_p99_ms": [f["p99_ms"] for f in floors]}


def main():
    build()
    out = {}
    only_stream = len(sys.argv) > 1 and sys.argv[1] == "stream"
    only_new = len(sys.argv) > 1 and sys.argv[1] == "new"          # only the keys the file does not hold yet
    if only_stream or only_new:
        with open(DIGESTS) as f:
            out = json.load(f)
    for wk in ([] if only_stream else WORKLOADS):
        mode, n, dist = wk[:3]
        seed = wk[3] if len(wk) > 3 else 1
        window = 25 if mode == "1v1" else 50
        if only_new and workload_key(mode, n, window, dist, seed) in out:
            continue
        key, d, matched = digest_of(mode, n, dist, seed)
        out[key] = d
        print(key, d, "matched", matched, flush=True)
    for label, seconds, qps in STREAMS:
        if only_new and stream_key(label, qps, seconds, 10.0) in out:
            continue
        key, rec = stream_digest_of(label, seconds, qps)
        out[key] = rec
        print(key, rec["digest"], "matched", rec["matched"], "backlog", [sum(b) for b in rec["backlog"]], flush=True)
    with open(DIGESTS, "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
        f.write("\n")


if __name__ == "__main__":
    main()
