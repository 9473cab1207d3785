#!/usr/bin/env python
"""tests/golden/literal_64k_digests.json: oracle/literal_ref.py — the line-by-line restatement of
lib/search/worker.ex:291-324, lib/models/lobby_state.ex:61-131, lib/requeue/worker.ex:51-54 — run ONCE
on pools of the size at which the HIP kernels branch into their long-chain paths (pair tiles from 16 384
players per chain, the team path from 4 096), chain by chain (docs/MATCH_CHECK.md section 4).

The small random scripts of tests/test_oracle_literal.py pin oracle/mode_r.c on pools of 20-160 players;
here the same literal code walks 65 536-player pools (minutes of pure Python per chain, hence committed
digests instead of a test that runs it): per (mode, rating group) and tick the emission list (publish
order, players in team order), the pass of every lobby, the stored lobby and the queue order (= requeue
order) after the tick, and the pair evaluations.  tests/test_oracle_literal_64k.py runs the C oracle on
the same seeded pools and compares; the GPU tier compares the engine with the same file.

TEST INFRASTRUCTURE.  The product never reads it.

    python tools/make_literal_digests.py [family ...]      # 8 processes, about ten minutes
"""
import hashlib
import json
import multiprocessing as mp
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from microservice_matchmaking_amd.config import (REFERENCE_RATING_GROUPS, make_config, mode_1v1,  # noqa: E402
                                                 mode_dicts, mode_team)
from microservice_matchmaking_amd.stream import stream_batch  # noqa: E402
from microservice_matchmaking_amd.synth import ROLE_WEIGHTS_5V5, make_pool  # noqa: E402
from oracle.literal_ref import SearchStage, find_rating_group_by_rating, team_name  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden", "literal_64k_digests.json")
OUT_1M = os.path.join(ROOT, "tests", "golden", "literal_1m_digests.json")
OUT_10M = os.path.join(ROOT, "tests", "golden", "literal_10m_digests.json")
N = 65536


def families():
    """name -> (modes, [script steps]).  A step is ("enqueue", n, seed, make_pool kwargs) |
    ("cancel", seed, one player in `every`) | ("tick",)."""
    return {
        # BASELINE cfg-2's mode on a 64k pool: the 0-1499 chain (19.6k players) is walked by kp_round tiles
        "1v1_w25_region": ([mode_1v1(window=25, region_filter=True)],
                           [("enqueue", N, 1, {}), ("tick",)]),
        # BASELINE cfg-3's mode: every chain above the team path's 4 096 players
        "5v5_w50_roles": ([mode_team(5, 2, 50, (1, 1, 1, 1, 1))],
                          [("enqueue", N, 1, {"role_weights": ROLE_WEIGHTS_5V5}), ("tick",)]),
        # the same two modes at 262 144 players: the longest 1v1 chain (78k players) starts at the longest tile length and
        # goes through all three, compactions included; the 5v5 chains run their lobby-rich passes
        "1v1_w25_region_256k": ([mode_1v1(window=25, region_filter=True)],
                                [("enqueue", 4 * N, 1, {}), ("tick",)]),
        "5v5_w50_roles_256k": ([mode_team(5, 2, 50, (1, 1, 1, 1, 1))],
                               [("enqueue", 4 * N, 1, {"role_weights": ROLE_WEIGHTS_5V5}), ("tick",)]),
        # BASELINE cfg-2 / cfg-3 THEMSELVES: the bench's 1M pools (seed 1).  The 0-1499 chain holds 300k players = 37 tiles of
        # 8192: the kp_round -> kp_ask_compact -> kp_rounds hand-over and the 7 chains x 32 tiles XCD map only exist here.
        # Written to tests/golden/literal_1m_digests.json (OUT_1M); the longest chain is most of an hour of CPython.
        "1v1_w25_region_1m": ([mode_1v1(window=25, region_filter=True)],
                              [("enqueue", 1000000, 1, {}), ("tick",)]),
        "5v5_w50_roles_1m": ([mode_team(5, 2, 50, (1, 1, 1, 1, 1))],
                             [("enqueue", 1000000, 1, {"role_weights": ROLE_WEIGHTS_5V5}), ("tick",)]),
        # BASELINE cfg-5's 70/30 mix at 1M players with cancels: tick, every 16th waiting player cancelled, tick (stale lobbies, purge,
        # heads that sit out), 131 072 late arrivals, every 32nd cancelled, tick — the cancel paths of the long-chain kernels
        # (kp_init's pos0 / extra0, kt_init's sit-out, k_purge over 300k-player queues) at the size the 1M families pin the plain tick
        "mixed_70_30_cancel_1m": ([mode_1v1(window=25, region_filter=True), mode_team(5, 2, 50, (1, 1, 1, 1, 1))],
                                  [("enqueue", 1000000, 1, {"role_weights": ROLE_WEIGHTS_5V5, "mode_weights": (70, 30)}),
                                   ("tick",), ("cancel", 7, 16), ("tick",),
                                   ("enqueue", 131072, 2, {"role_weights": ROLE_WEIGHTS_5V5, "mode_weights": (70, 30)}),
                                   ("cancel", 8, 32), ("tick",)]),
        # BASELINE cfg-4's pool (10M players, the shared pool of bench.py --gpus N and of its shared_pool_n1 leg): the 0-1499
        # chain holds 3.0M players = 367 tiles of 8192 — the second level of the route (kp_group, chains of 64+ tiles) and the
        # two rounds of workgroups only exist here.  Hours of CPython; tests/golden/literal_10m_digests.json (OUT_10M).
        "1v1_w25_region_10m": ([mode_1v1(window=25, region_filter=True)],
                               [("enqueue", 10000000, 1, {}), ("tick",)]),
        # BASELINE cfg-5's mix in one pool, a cancel tick (stale lobbies, purge) and late arrivals
        "mixed_70_30_cancel": ([mode_1v1(window=25, region_filter=True), mode_team(5, 2, 50, (1, 1, 1, 1, 1))],
                               [("enqueue", N, 1, {"role_weights": ROLE_WEIGHTS_5V5, "mode_weights": (70, 30)}),
                                ("tick",), ("cancel", 7, 16), ("tick",),
                                ("enqueue", 8192, 2, {"role_weights": ROLE_WEIGHTS_5V5, "mode_weights": (70, 30)}),
                                ("cancel", 8, 32), ("tick",)]),
    }


def payload(idx, rating, cons):
    c = int(cons)
    return {"id": int(idx), "rating": int(rating), "game-mode": "mode%d" % (c & 0xF),
            "region": (c >> 4) & 0xFF, "party": (c >> 12) & 0xF, "role": (c >> 16) & 0xF}


def script_arrays(steps):
    """The arrival batches of a script as (first global index, rating, cons)."""
    out, first = [], 0
    for st in steps:
        if st[0] == "enqueue":
            if "mode_weights" in st[3]:           # players of the 1v1 mode carry no role (stream.stream_batch)
                rating, cons = stream_batch(st[1], st[2], st[3]["mode_weights"], st[3].get("role_weights"))
            else:
                rating, cons = make_pool(st[1], seed=st[2], **st[3])
            out.append((first, rating, cons))
            first += st[1]
    return out


def cancel_ids(seed, every, alive_ids):
    """Which of the players still waiting are cancelled: a seeded choice that needs no RNG state shared
    with anybody (the C oracle side of the test computes the same set from the same list)."""
    ids = np.asarray(sorted(alive_ids), dtype=np.uint64)
    with np.errstate(over="ignore"):
        h = (ids * np.uint64(0x9E3779B97F4A7C15) + np.uint64(seed) * np.uint64(0xBF58476D1CE4E5B9)) >> np.uint64(40)
    return [int(i) for i in ids[(h % np.uint64(every)) == 0]]


def h16(arr):
    return hashlib.blake2b(np.ascontiguousarray(arr, dtype="<i8").tobytes(), digest_size=16).hexdigest()


def run_chain(task):
    """One (family, mode, group): the literal stage fed this chain's players only."""
    fam, mode, gi = task
    modes, steps = families()[fam]
    cfg = make_config(modes, capacity=1 << 19, timing=False)
    md = mode_dicts(cfg)
    stage = SearchStage({"mode%d" % i: m for i, m in enumerate(md)}, REFERENCE_RATING_GROUPS)
    gname = REFERENCE_RATING_GROUPS[gi][2]
    t0 = time.time()
    batches = iter(script_arrays(steps))
    waiting = set()
    ticks = []
    for st in steps:
        if st[0] == "enqueue":
            first, rating, cons = next(batches)
            for k in range(len(rating)):
                if (int(cons[k]) & 0xF) != mode:
                    continue
                if find_rating_group_by_rating(int(rating[k]), REFERENCE_RATING_GROUPS)[2] != gname:
                    continue
                stage.deliver(payload(first + k, rating[k], cons[k]))
                waiting.add(first + k)
        elif st[0] == "cancel":
            # the cancel set is drawn over ALL chains' waiting players; this chain sees its own
            for i in cancel_ids(st[1], st[2], waiting):
                stage.cancel(i)
                waiting.discard(i)
        else:
            n0, p0 = len(stage.emitted), stage.pairs
            plog = []
            stage.run_group_to_quiescence(gname, pass_log=plog)
            em = stage.emitted[n0:]
            teams = cfg.modes[mode].teams
            ids = [p["id"] for e in em for t in range(teams) for p in e["teams"][team_name(t)]]
            waiting.difference_update(ids)
            lobby = []
            for rec in stage.lobbies.tables[gname]:
                if rec[2] == "mode%d" % mode:
                    lobby = [p["id"] for t in range(teams) for p in rec[1].get(team_name(t), [])]
            queue = [p["id"] for p in stage.queues[gname] if stage.active.in_queue(p["id"])]
            ticks.append({"lobbies": len(em), "emission": h16(ids), "passes": h16(plog), "lobby": lobby,
                          "queue_len": len(queue), "queue": h16(queue), "pairs": stage.pairs - p0})
    return fam, mode, gi, ticks, time.time() - t0


def main():
    want = sys.argv[1:] or [f for f in families() if not f.endswith(("_1m", "_10m"))]
    out_path = OUT
    for suffix, path in (("_1m", OUT_1M), ("_10m", OUT_10M)):
        if any(f.endswith(suffix) for f in want):
            assert all(f.endswith(suffix) for f in want), "the %s families go to their own file: run them alone" % suffix
            out_path = path
    tasks = []
    for fam in want:
        modes, _ = families()[fam]
        for mode in range(len(modes)):
            for gi in range(len(REFERENCE_RATING_GROUPS)):
                tasks.append((fam, mode, gi))
    tasks.sort(key=lambda t: {0: 0, 6: 1}.get(t[2], 2))     # the two wide groups first: they are the wall time
    out = {}
    if os.path.exists(out_path):
        out = json.load(open(out_path))
    with mp.Pool(min(8, os.cpu_count() or 1)) as pool:
        for fam, mode, gi, ticks, dt in pool.imap_unordered(run_chain, tasks):
            out.setdefault(fam, {})["%d/%d" % (mode, gi)] = ticks
            for t in ticks:
                t["literal_seconds"] = round(dt)
            print("%s mode %d group %d: %s lobbies, %.0f s" % (fam, mode, gi, [t["lobbies"] for t in ticks], dt), flush=True)
            if out_path != OUT:               # hours of CPU: keep what is done
                json.dump(out, open(out_path + ".part", "w"), indent=1, sort_keys=True)
    out["_about"] = ("oracle/literal_ref.py on %s pools, chain by chain; tools/make_literal_digests.py; "
                     "keys family -> 'mode/group' -> one record per tick"
                     % ("BASELINE cfg-2 / cfg-3's 1 000 000-player" if out_path == OUT_1M else
                        "BASELINE cfg-4's 10 000 000-player" if out_path == OUT_10M else "65 536-player"))
    with open(out_path, "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
        f.write("\n")


if __name__ == "__main__":
    main()
