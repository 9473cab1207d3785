"""Shared drivers: run a golden case or a random scenario against any ABI engine class."""
from __future__ import annotations

import json
import os

import numpy as np

from microservice_matchmaking_amd._abi import NO_SLOT, cons_make
from microservice_matchmaking_amd.config import make_config

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "mode_r_cases.json")


def load_golden():
    with open(GOLDEN) as f:
        return json.load(f)


def players_to_arrays(players):
    p = np.asarray(players, dtype=np.int64).reshape(-1, 5)
    rating = p[:, 0].astype(np.int32)
    cons = cons_make(p[:, 4], p[:, 1], p[:, 2], p[:, 3])
    return rating, cons


def run_golden_case(engine_cls, case, capacity=64):
    cfg = make_config(case["modes"], capacity=capacity)
    eng = engine_cls(cfg)
    try:
        for step in case["steps"]:
            if step["op"] == "enqueue":
                r, c = players_to_arrays(step["players"])
                slots = eng.enqueue(r, c)
                if "expect_slots" in step:
                    assert slots.tolist() == step["expect_slots"], (case["name"], slots)
            elif step["op"] == "cancel":
                eng.cancel(np.asarray(step["slots"], dtype=np.uint32))
            elif step["op"] == "tick":
                m = eng.tick(step["mode"])
                ex = step["expect"]
                tag = "%s tick(mode %d)" % (case["name"], step["mode"])
                assert m.slots.tolist() == ex["matches"], (tag, m.slots.tolist())
                assert m.group.tolist() == ex["group"], tag
                assert m.pass_.tolist() == ex["pass"], tag
                assert np.allclose(m.score, np.asarray(ex["score"], np.float32), atol=1e-6), tag
                assert m.stats["pairs"] == ex["pairs"], (tag, m.stats["pairs"])
                assert m.stats["passes_max"] == ex["passes_max"], (tag, m.stats["passes_max"])
                assert m.stats["pool_after"] == ex["pool_after"], (tag, m.stats["pool_after"])
                assert m.stats["matches"] == len(ex["matches"]), tag
                assert eng.queue_depth(step["mode"]).tolist() == ex["depth"], (tag, eng.queue_depth(step["mode"]))
                for g, lb in ex["lobby"].items():
                    s, t = eng.lobby_state(step["mode"], int(g))
                    assert s.tolist() == lb["slots"], (tag, g, s)
                    assert t.tolist() == lb["teams"], (tag, g, t)
            else:
                raise ValueError(step["op"])
    finally:
        eng.close()


def assert_same_tick(a, b, tag="", score_tol=1e-6):
    """a, b: Matches.  Bit-exact assignments/order; scores within tolerance."""
    assert a.slots.shape == b.slots.shape, (tag, a.slots.shape, b.slots.shape)
    assert np.array_equal(a.slots, b.slots), (tag, "slots differ at row",
                                              int(np.argmax((a.slots != b.slots).any(axis=1))))
    assert np.array_equal(a.group, b.group), (tag, "group")
    assert np.array_equal(a.pass_, b.pass_), (tag, "pass")
    assert np.allclose(a.score, b.score, atol=score_tol, rtol=0), (tag, "score")
    for k in ("pool_before", "pool_after", "matches", "players_matched", "passes_max", "pairs", "scanned"):
        assert a.stats[k] == b.stats[k], (tag, k, a.stats[k], b.stats[k])


def assert_same_state(ea, eb, cfg, tag=""):
    """Queue depth, queue ORDER (requeue ordering, worker.ex:239-248 -> requeue/worker.ex:51-54: the
    survivors of a tick in the order the broker would deliver them next) and the stored lobby."""
    for mode in range(cfg.n_modes):
        assert np.array_equal(ea.queue_depth(mode), eb.queue_depth(mode)), (tag, "depth", mode)
        for g in range(cfg.n_groups):
            qa, qb = ea.queue_slots(mode, g), eb.queue_slots(mode, g)
            assert np.array_equal(qa, qb), (tag, "queue order", mode, g,
                                            int(np.argmax(qa != qb)) if qa.shape == qb.shape else (qa.shape, qb.shape))
            sa, ta = ea.lobby_state(mode, g)
            sb, tb = eb.lobby_state(mode, g)
            assert np.array_equal(sa, sb) and np.array_equal(ta, tb), (tag, "lobby", mode, g, sa, sb)


def random_scenario(rng, cfg, ea, eb, n_rounds=4, batch=200, cancel_frac=0.05, n_regions=3,
                    rating_lo=0, rating_hi=5000, n_parties=2):
    """Drive two engines with the same random enqueue/cancel/tick script and compare."""
    live = []
    for rnd in range(n_rounds):
        n = int(rng.integers(0, batch + 1))
        rating = rng.integers(rating_lo, rating_hi + 1, size=n).astype(np.int32)
        mode = rng.integers(0, cfg.n_modes, size=n)
        role = np.array([rng.integers(0, cfg.modes[int(m)].n_roles) for m in mode], dtype=np.uint32)
        region = rng.integers(0, n_regions, size=n)
        party = rng.integers(0, n_parties, size=n)
        cons = cons_make(mode, region, party, role)
        sa = ea.enqueue(rating, cons)
        sb = eb.enqueue(rating, cons)
        assert np.array_equal(sa, sb), "slots"
        live.extend(int(s) for s in sa if s != NO_SLOT)
        if live and cancel_frac > 0:
            k = int(len(live) * cancel_frac)
            if k:
                idx = rng.choice(len(live), size=k, replace=False)
                cs = np.asarray([live[i] for i in idx], dtype=np.uint32)
                ea.cancel(cs)
                eb.cancel(cs)
                gone = set(cs.tolist())
                live = [s for s in live if s not in gone]
        for mode_i in range(cfg.n_modes):
            ma = ea.tick(mode_i)
            mb = eb.tick(mode_i)
            assert_same_tick(ma, mb, tag="round %d mode %d" % (rnd, mode_i))
            gone = set(ma.slots.ravel().tolist())
            live = [s for s in live if s not in gone]
        assert_same_state(ea, eb, cfg, tag="round %d" % rnd)


def run_wrapping_stream(engine_cls, oracle_cls, capacity=4096, ticks=120, per_tick=500, seed=17):
    """A stream that laps the slot ring many times while some players wait for the whole run
    (loners in thinly populated rating groups and regions): the handles of every batch, every
    tick and the final state must be those of the oracle, and no batch may be refused while
    the pool has free slots.  Returns (laps of the ring, batches that had to step over a slot)."""
    from microservice_matchmaking_amd.config import mode_1v1
    cfg = make_config([mode_1v1(window=25, region_filter=True)], capacity=capacity)
    rng = np.random.default_rng(seed)
    handed, stepped = 0, 0
    with engine_cls(cfg) as a, oracle_cls(cfg) as b:
        for t in range(ticks):
            n = int(rng.integers(per_tick // 2, per_tick + 1))
            # most of the traffic in one dense band; a few loners in the other rating groups (a loner
            # inside the band's own group would become an anchor nobody fits and hold the whole chain
            # up until a fitting player arrives: reference behaviour, MATCH_CHECK.md section 4)
            rating = np.where(rng.random(n) < 0.97, rng.integers(2000, 2100, size=n),
                              rng.choice([rng.integers(0, 2000), rng.integers(2500, 5001)], size=n)).astype(np.int32)
            cons = cons_make(0, rng.integers(0, 4, size=n), 0, 0)
            sa, sb = a.enqueue(rating, cons), b.enqueue(rating, cons)
            assert np.array_equal(sa, sb), ("handles differ at tick", t)
            handed += n
            stepped += int((np.diff(sa.astype(np.int64)) % capacity != 1).any())
            if t % 7 == 3:                                            # some of the waiting give up
                depth_before = int(a.queue_depth(0).sum())
                if depth_before:
                    cs = rng.choice(sa, size=min(5, n), replace=False)
                    a.cancel(cs)
                    b.cancel(cs)
            ma, mb = a.tick(0), b.tick(0)
            assert_same_tick(ma, mb, "wrapping stream tick %d" % t)
        assert_same_state(a, b, cfg, "wrapping stream")
    return handed / capacity, stepped


def run_starving_team_stream(engine_cls, oracle_cls, preload=200_000, ticks=40, per_tick=400, cancels=25, seed=11,
                             capacity=1 << 19):
    """cfg-5 where it ends up after a minute (reference lib/search/worker.ex:352-358 deliveries, :291-324 attempts):
    cfg-3's role weights give 10 % supports for 20 % of the seats, so half of the 5v5 arrivals can never be seated and
    the pool grows into chains of tens of thousands of players in which a tick seats a handful of lobbies — the team
    path (chains >= 4096 players) in its starving regime, every tick, with cancels trickling in.  Engine vs oracle,
    after every tick: lobbies, order, counters, queue order, stored lobby.  Returns the lobbies per tick."""
    from microservice_matchmaking_amd.config import mode_team
    from microservice_matchmaking_amd.synth import ROLE_WEIGHTS_5V5, make_pool
    cfg = make_config([mode_team(5, 2, 50, (1, 1, 1, 1, 1))], capacity=capacity)
    rng = np.random.default_rng(seed)
    per = []
    with engine_cls(cfg) as a, oracle_cls(cfg) as b:
        rating, cons = make_pool(preload, seed=seed, role_weights=ROLE_WEIGHTS_5V5)
        sa, sb = a.enqueue(rating, cons), b.enqueue(rating, cons)
        assert np.array_equal(sa, sb)
        waiting = set(sa.tolist())
        ma, mb = a.tick(0), b.tick(0)                      # the backlog a minute of the stream leaves behind
        assert_same_tick(ma, mb, "starving stream: preload")
        waiting -= set(ma.slots.ravel().tolist())
        depth = a.queue_depth(0)
        for t in range(ticks):
            rating, cons = make_pool(per_tick, seed=1000 * seed + t, role_weights=ROLE_WEIGHTS_5V5)
            sa, sb = a.enqueue(rating, cons), b.enqueue(rating, cons)
            assert np.array_equal(sa, sb), ("handles differ at tick", t)
            waiting |= set(sa.tolist())
            if cancels and t % 2 == 1:
                cs = rng.choice(np.fromiter(waiting, dtype=np.uint32), size=cancels, replace=False)
                a.cancel(cs)
                b.cancel(cs)
                waiting -= set(cs.tolist())
            ma, mb = a.tick(0), b.tick(0)
            assert_same_tick(ma, mb, "starving stream tick %d" % t)
            assert_same_state(a, b, cfg, "starving stream tick %d" % t)
            waiting -= set(ma.slots.ravel().tolist())
            per.append(len(ma))
    return per, depth
