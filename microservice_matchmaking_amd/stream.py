"""The streaming leg of the path (BASELINE cfg-5): a Poisson arrival stream, one enqueue and one
tick per mode every tick period, on one engine or on one rank's share of the chains.

What replaces what: the arrivals of a period are the deliveries `Search.Worker` would receive
(reference lib/search/worker.ex:352-358), a tick is `consume/5` run to quiescence for one mode
(worker.ex:291-324).  The stream is driven by TICK COUNT, not by wall time, so what is matched
in which tick is deterministic (and is compared with the oracle in tests/); wall time only
enters the reported latency.

Two latencies per matched player:
  real   wall time at which its lobby came back from mm_tick, minus its arrival time;
  floor  end of the tick period in which it was matched, minus its arrival time — what an
         engine with a free tick would give.  It is the time the player waited for fitting
         partners to ARRIVE (reference behaviour: a lobby waits until a fitting player is
         delivered, docs/MATCH_CHECK.md §4); real - floor is what the engine adds.
"""
from __future__ import annotations

import hashlib
import time

import numpy as np

from ._abi import MMError
from .synth import make_pool


def stream_batch(n, seed, mode_weights=None, role_weights=None):
    """One period's arrivals.  Players of mode 0 (1v1) carry no role."""
    rating, cons = make_pool(n, seed=seed, mode_weights=mode_weights, role_weights=role_weights)
    if mode_weights:
        cons = np.where((cons & 0xF) == 0, cons & ~np.uint32(0xF << 16), cons).astype(np.uint32)
    return rating, cons


def stream_schedule(qps, seconds, tick_ms, seed):
    """[(t_open, t_close, n, batch seed, sorted arrival times)] — the same on every rank."""
    rng = np.random.default_rng(seed)
    n_ticks = int(seconds * 1000.0 / tick_ms)
    out = []
    for k in range(n_ticks):
        t_open, t_close = k * tick_ms * 1e-3, (k + 1) * tick_ms * 1e-3
        n = int(rng.poisson(qps * tick_ms * 1e-3))
        out.append((t_open, t_close, n, seed + 1 + k, np.sort(rng.uniform(t_open, t_close, size=n))))
    return out


def run_stream(search, schedule, mode_weights=None, role_weights=None, realtime=True, batches=None):
    """Drive `search` (a sharding.ShardedSearch: one engine + the chains this rank owns) through
    the schedule.  Returns a dict with per-mode latency arrays (real, floor), matched players,
    per-tick cost and a digest per chain of everything it emitted, in order.
    `batches`: the arrivals of every tick of the schedule, made by the caller before the clock starts
    (stream_batch(n, seed, ...) per entry — at 200 000 players a tick numpy needs longer than the 10 ms period to
    draw them, and a leg would fall behind because of the host, not the engine)."""
    assert batches is None or len(batches) == len(schedule)
    cfg = search.cfg
    n_modes, n_groups = int(cfg.n_modes), int(cfg.n_groups)
    total = sum(s[2] for s in schedule)
    arrival = np.zeros(total, dtype=np.float64)                 # by global arrival index
    real = [[] for _ in range(n_modes)]
    floor = [[] for _ in range(n_modes)]
    hashers = {(m, g): hashlib.blake2b(digest_size=16) for m in range(n_modes) for g in range(n_groups)}
    emitted = {key: 0 for key in hashers}
    tick_cost = []
    matched = 0
    first = 0
    full_at_s = None
    # (a generation-2 collection of the interpreter's heap is a pause of tens of milliseconds in one tick of a real-time
    # run: nothing here makes reference cycles, so the collector rests until the stream is over)
    import gc
    gc_was = gc.isenabled()
    if realtime:
        gc.disable()
    t_start = time.perf_counter()
    try:
        for k, (t_open, t_close, n, sd, ts) in enumerate(schedule):
            rating, cons = batches[k] if batches is not None else stream_batch(n, sd, mode_weights, role_weights)
            arrival[first:first + n] = ts
            if realtime:
                # the period has to be over: sleep through most of it and spin only for the last stretch (a thread that spins
                # all the time is the first one a CPU quota throttles, for tens of milliseconds at a time)
                while True:
                    left = t_close - (time.perf_counter() - t_start)
                    if left <= 0:
                        break
                    if left > 0.0006:
                        time.sleep(left - 0.0004)
            t0 = time.perf_counter()
            try:
                search.enqueue(rating, cons, first_global_index=first)
            except MMError as ex:
                if ex.status != -4:                                 # MM_ERR_FULL: fewer than n free slots in the pool
                    raise
                # The batch is refused as a whole and nothing of it was queued (include/mm_engine.h): for the
                # service these deliveries stay unacked in the broker (prefetch back-pressure,
                # lib/search/worker.ex:29) until lobbies free slots.  The stream ends here and says so.
                full_at_s = t_open
                break
            first += n
            for md in range(n_modes):
                m = search.tick(md)
                t1 = time.perf_counter()
                if len(m):
                    ids = search.global_ids(m)
                    flat = ids.ravel()
                    real[md].append((t1 - t_start) - arrival[flat])
                    floor[md].append(t_close - arrival[flat])
                    matched += flat.size
                    for g in np.unique(m.group):
                        sel = np.ascontiguousarray(ids[m.group == g], dtype="<i8")
                        hashers[(md, int(g))].update(sel.tobytes())
                        emitted[(md, int(g))] += int(sel.shape[0])
            tick_cost.append(time.perf_counter() - t0)
    finally:
        # whatever ends the loop (a refused batch is handled above; a failed tick, Ctrl-C): the interpreter gets its collector back
        if realtime and gc_was:
            gc.enable()
    elapsed = time.perf_counter() - t_start
    depth = [search.engine.queue_depth(md).astype(np.int64) for md in range(n_modes)]
    cat = lambda parts: np.concatenate(parts) if parts else np.zeros(0)
    return {
        "real": [cat(x) for x in real], "floor": [cat(x) for x in floor],
        "matched": matched, "elapsed": elapsed, "tick_cost": np.asarray(tick_cost),
        "depth": depth, "digests": {k: h.hexdigest() for k, h in hashers.items()}, "lobbies": emitted,
        "arrivals": total, "ingested": first, "full_at_s": full_at_s,
    }


def latency_summary(real, floor):
    """p50 / p99 / max of the real latency and of the arrival-limited floor, in ms."""
    if real.size == 0:
        return {"p50_ms": None, "p99_ms": None, "max_ms": None, "floor_p50_ms": None, "floor_p99_ms": None,
                "matched_players": 0}
    return {"p50_ms": float(np.percentile(real, 50) * 1e3), "p99_ms": float(np.percentile(real, 99) * 1e3),
            "max_ms": float(real.max() * 1e3),
            "floor_p50_ms": float(np.percentile(floor, 50) * 1e3), "floor_p99_ms": float(np.percentile(floor, 99) * 1e3),
            "engine_added_p99_ms": float(np.percentile(real - floor, 99) * 1e3),
            "matched_players": int(real.size)}
