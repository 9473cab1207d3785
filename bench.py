#!/usr/bin/env python
"""bench.py — matched players/sec over a 1M-player pool (BASELINE.json metric), on N MI355X.

One "step" = one pass of the hot path over one batch of synthetic input: reset the engine,
enqueue a 1,000,000-player pool that is ALREADY RESIDENT IN HBM (bucketing kernels), run
the search to quiescence (walk kernel), and bring the match list back to the host (the
service needs it to publish lobbies).  Workload = BASELINE.json configs[1]:
"1v1, 1M players, +-25 rating + region filter on 1 MI355X", seeded synthetic pool
(uniform integer ratings on [0, 5000], 8 regions, arrival order = index).

N > 1: rating-group chains are independent (reference lib/application.ex:26-40), so the
path shards with no data-path collective; every rank owns one engine and its own 1M-player
pool (another seed) — weak scaling.  RCCL is used only for the barrier and the max/sum of
the timing/throughput scalars.

Prints ONE JSON line (rank 0).  `roofline` is for the walk (the search proper: for a 1v1
mode the pair path's kernel sequence kp_nx_init, kp_round x rounds (one launch per pass of the
cursor), kp_late, kp_finish — DESIGN.md §4; for team modes the single kernel k_walk):
achieved = algorithmic bytes / HIP-event time of that sequence on the engine's stream,
algorithmic bytes = pair evaluations of the reference algorithm (the oracle's count, which
the engine reproduces bit-exactly) x 8 B (rating + cons of the candidate).
The engine does NOT touch every such pair — it keeps next[] pointers instead of rescanning —
so `traffic` (HBM bytes from rocprofv3 PMC passes, profiles/) is far below the algorithmic
bytes; the walk is bound by the latency of ~300-600 dependent passes, not by HBM
(DESIGN.md §5).  `cpu_baseline` = the oracle
(oracle/mode_r.c, a C port of the reference's sequential search, in-memory — an upper
bound on what the BEAM service could do) timed on this box's host cores.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec
HBM_COPY_GBS = 6290.0          # MI355X_MICROARCH.md: measured copy ceiling (SURVEY.md §8(d) asks for both)
BYTES_PER_PAIR = 8


def roofline_block(mode, pairs, bytes_per_pair, walk_ms, step_ms, players, passes_max, traffic):
    """The `roofline` object of the bench line (pure arithmetic; tests/test_bench_line.py).
    SURVEY.md §8(d): achieved = algorithmic bytes / walk time, with its mandatory companions —
    (i) physical HBM GB/s (PMC traffic / walk time), (ii) the compulsory bytes of a tick
    (every player read once, 12 B, and written out once, 8 B) and how close the whole step is
    to streaming just those, (iii) the tile width of the dominant kernel."""
    walk_name = ("pair walk: kp_nx_init + kp_round x rounds + kp_late + kp_finish"
                 if mode == "1v1" else "team walk: (kt_build + kt_f + kt_chase + kt_emit) x passes")
    achieved = pairs * bytes_per_pair / (walk_ms * 1e-3) / 1e9 if walk_ms > 0 else 0.0
    compulsory = float(players) * (12 + 8)
    return {
        "bound": "hbm", "kernel": walk_name, "achieved": achieved, "peak": HBM_PEAK_GBS,
        "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
        "algorithmic_bytes_per_launch": pairs * bytes_per_pair,
        "launch": "one tick = one walk to quiescence (HIP events around the kernel sequence)",
        "frac_of_measured_copy_peak": achieved / HBM_COPY_GBS,
        "physical_gbs": (traffic / (walk_ms * 1e-3) / 1e9) if (traffic and walk_ms > 0) else None,
        "compulsory_bytes_per_tick": compulsory,
        "compulsory_frac": (compulsory / (step_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if step_ms > 0 else None,
        "tile_positions": 8192 if mode == "1v1" else 512,
        "note": ("Mode R is a chain of dependent first-fit steps (%d passes of the cursor for "
                 "the longest rating group); the engine replaces the per-pair rescans by "
                 "%s, so it is bound by pass latency (kernel boundaries + LDS/VALU issue), "
                 "not by HBM; see DESIGN.md") % (
                    passes_max,
                    "next[] pointers repaired incrementally" if mode == "1v1" else
                    "per-pass F pointers (who the cursor picks after each player's lobby) "
                    "computed for every queued player at once"),
    }


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--players", type=int, default=1_000_000)
    ap.add_argument("--window", type=int, default=25)
    ap.add_argument("--dist", default="uniform", choices=["uniform", "normal"])
    ap.add_argument("--mode", default="1v1", choices=["1v1", "5v5"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-seconds", type=float, default=20.0, help="CPU time budget of the cpu_baseline leg")
    ap.add_argument("--no-stream", action="store_true", help="skip the streaming-latency leg")
    ap.add_argument("--concurrent-pools", type=int, default=0,
                    help="opt-in extra leg (N=1 only): this many engines, each with its own pool of --players, "
                         "ticking concurrently on their own streams; reported beside the main line, never as `value`")
    ap.add_argument("--stream-qps", type=int, default=100_000)
    ap.add_argument("--stream-seconds", type=float, default=3.0)
    ap.add_argument("--stream-tick-ms", type=float, default=10.0)
    ap.add_argument("--traffic-json", default=None,
                    help="optional {'k_walk_hbm_bytes_per_launch': ...} from a rocprofv3 --pmc pass")
    return ap.parse_args()


def cpu_baseline(cfg, rating, cons, mode_name, budget_s=20.0):
    """The oracle on the same workload, on this host.  Bounded: whole pools until ~budget_s/2
    per variant (>= 2 repetitions).  Only the search (mo_tick) is timed — enqueue excluded,
    matching the reference where bucketing is a separate stage."""
    from oracle.oracle import OracleEngine
    out = {}
    for label, threads in (("1thread", 1), ("7threads", 7)):
        times, matched, reps = [], 0, 0
        t_start = time.time()
        while reps < 2 or (time.time() - t_start < budget_s / 2 and reps < 20):
            eng = OracleEngine(cfg)
            eng.enqueue(rating, cons)
            t0 = time.perf_counter()
            m = eng.tick(0) if threads == 1 else eng.tick_threads(0, threads)
            times.append(time.perf_counter() - t0)
            matched = m.stats["players_matched"]
            pairs = m.stats["pairs"]
            eng.close()
            reps += 1
        best = min(times)
        out[label] = {"players_per_s": matched / best, "pairs_per_s": pairs / best,
                      "ms": best * 1e3, "reps": reps}
    return {
        "value": out["1thread"]["players_per_s"], "unit": "matched players/s", "cores": 1,
        "kind": "port",
        "sample": "whole %d-player %s pool (search only, in-memory, no AMQP/Mnesia/JSON), best of %d runs"
                  % (len(rating), mode_name, out["1thread"]["reps"]),
        "ms": out["1thread"]["ms"], "pairs_per_s": out["1thread"]["pairs_per_s"],
        "threads7": {"value": out["7threads"]["players_per_s"], "cores": 7, "ms": out["7threads"]["ms"],
                     "note": "one thread per rating group = the reference's own parallelism"},
        "host_cores": os.cpu_count(),
    }


def concurrent_pools(make_engine, pools, steps, make_inputs):
    """Extra leg: the walk is bound by the latency of its passes and keeps well under half of the CUs
    busy, so independent pools (other regions / shards of a service) on engines of their own
    (own stream, own device memory: include/mm_engine.h) should overlap.  `pools` host threads, one
    engine each, `steps` steps each (the ctypes calls release the GIL); matched players of all
    pools over the wall time of the slowest."""
    import threading
    engines = [make_engine() for _ in range(pools)]
    inputs = [make_inputs(k) for k in range(pools)]
    matched = [0] * pools
    errors = []
    start = threading.Barrier(pools + 1)

    def work(k):
        try:
            eng, (d_rating, d_cons) = engines[k], inputs[k]
            eng.reset()
            eng.enqueue_device(d_rating, d_cons)
            eng.tick(0)                                   # warm-up
            start.wait()
            for _ in range(steps):
                eng.reset()
                eng.enqueue_device(d_rating, d_cons)
                matched[k] += int(eng.tick(0).stats["players_matched"])
        except Exception as ex:                           # report, never hang the barrier
            errors.append(repr(ex))
            start.abort()

    threads = [threading.Thread(target=work, args=(k,)) for k in range(pools)]
    for t in threads:
        t.start()
    try:
        start.wait()
    except threading.BrokenBarrierError:
        pass
    t0 = time.perf_counter()
    for t in threads:
        t.join()
    elapsed = time.perf_counter() - t0
    for e in engines:
        e.close()
    if errors:
        return {"pools": pools, "error": errors[0]}
    return {"pools": pools, "steps": steps, "value": sum(matched) / elapsed, "unit": "matched players/s",
            "ms_per_step_per_pool": elapsed / steps * 1e3,
            "note": "k independent pools on one GPU, one engine and stream each; not the headline workload"}


def stream_latency(make_engine, qps, seconds, tick_ms, label, seed=77, mode_weights=None, role_weights=None):
    """Second half of BASELINE.json's metric: match latency at a fixed enqueue rate.  Players
    arrive as a Poisson stream (`qps`), every `tick_ms` of REAL time the arrivals of the period
    are enqueued (host pointers, so the H2D copy and the bucketing kernels are inside) and every
    mode is ticked once; a matched player's latency = wall time at which its lobby came back
    from mm_tick minus its arrival time.  Reported: p50 / p99 / max, matched players/s, backlog.
    `mode_weights` mixes the engine's modes (BASELINE cfg-5: 70 % 1v1 / 30 % 5v5); players of
    mode 1 draw a role from `role_weights`."""
    import time as _t
    from microservice_matchmaking_amd.synth import make_pool
    eng = make_engine()
    n_modes = int(eng.cfg.n_modes)
    rng = np.random.default_rng(seed)
    n_ticks = int(seconds * 1000.0 / tick_ms)
    arrival = np.zeros(eng.cfg.capacity, dtype=np.float64)     # by slot
    lat = [[] for _ in range(n_modes)]
    tick_cost = []
    matched = 0

    def batch(n, sd):
        rating, cons = make_pool(n, seed=sd, mode_weights=mode_weights, role_weights=role_weights)
        if mode_weights:
            cons = np.where((cons & 0xF) == 0, cons & ~np.uint32(0xF << 16), cons).astype(np.uint32)
        return rating, cons

    # warm the kernels up outside the clock
    r0, c0 = batch(2000, seed)
    eng.enqueue(r0, c0)
    for md in range(n_modes):
        eng.tick(md)
    eng.reset()
    t_start = _t.perf_counter()
    for k in range(n_ticks):
        t_open, t_close = k * tick_ms * 1e-3, (k + 1) * tick_ms * 1e-3
        n = int(rng.poisson(qps * tick_ms * 1e-3))
        rating, cons = batch(n, seed + 1 + k)
        ts = np.sort(rng.uniform(t_open, t_close, size=n))
        while _t.perf_counter() - t_start < t_close:            # the period has to be over
            pass
        t0 = _t.perf_counter()
        slots = eng.enqueue(rating, cons)
        arrival[slots] = ts
        for md in range(n_modes):
            m = eng.tick(md)
            t1 = _t.perf_counter()
            if len(m):
                s_ = m.slots.ravel()
                lat[md].append((t1 - t_start) - arrival[s_])
                matched += s_.size
        tick_cost.append(_t.perf_counter() - t0)
    elapsed = _t.perf_counter() - t_start
    depth = [int(eng.queue_depth(md).sum()) for md in range(n_modes)]
    eng.close()
    per_mode = []
    for md in range(n_modes):
        v = np.concatenate(lat[md]) if lat[md] else np.zeros(1)
        per_mode.append({"p50_ms": float(np.percentile(v, 50) * 1e3), "p99_ms": float(np.percentile(v, 99) * 1e3),
                         "max_ms": float(v.max() * 1e3), "matched_players": int(v.size if lat[md] else 0),
                         "backlog_players": depth[md]})
    allv = np.concatenate([np.concatenate(x) for x in lat if x]) if any(lat) else np.zeros(1)
    out = {"enqueue_qps": qps, "tick_ms": tick_ms, "seconds": seconds, "mode": label,
           "p50_ms": float(np.percentile(allv, 50) * 1e3), "p99_ms": float(np.percentile(allv, 99) * 1e3),
           "max_ms": float(allv.max() * 1e3), "matched_players_per_s": matched / elapsed,
           "tick_cost_ms_mean": float(np.mean(tick_cost) * 1e3), "tick_cost_ms_p99": float(np.percentile(tick_cost, 99) * 1e3),
           "backlog_players": int(sum(depth)), "kept_up": bool(elapsed < seconds * 1.05),
           "note": "latency floor = half a tick period + the tick; a chain whose anchor nobody fits waits for "
                   "arrivals (reference behaviour, docs/MATCH_CHECK.md section 4)"}
    if n_modes > 1:
        out["per_mode"] = per_mode
    return out


def main():
    args = parse()
    import torch
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert torch.cuda.is_available(), "bench.py needs a GPU (there is no CPU fallback)"
    torch.cuda.set_device(local_rank)
    dist_on = world > 1
    if dist_on:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    from microservice_matchmaking_amd import Engine, make_config, mode_1v1, mode_team
    from microservice_matchmaking_amd.synth import ROLE_WEIGHTS_5V5, make_pool

    n = args.players
    if args.mode == "1v1":
        modes = [mode_1v1(window=args.window, region_filter=True)]
        rating, cons = make_pool(n, seed=1 + rank, dist=args.dist)
        bytes_per_pair = BYTES_PER_PAIR
        workload = "1v1, %d players, +-%d rating + region filter (8 regions), %s ratings" % (n, args.window, args.dist)
    else:
        modes = [mode_team(5, 2, 50, (1, 1, 1, 1, 1))]
        rating, cons = make_pool(n, seed=1 + rank, dist=args.dist, role_weights=ROLE_WEIGHTS_5V5)
        bytes_per_pair = 12
        workload = "5v5 team balance, %d players, +-50 rating + 5 roles, %s ratings" % (n, args.dist)
    cap = 1
    while cap < n:
        cap <<= 1
    cfg = make_config(modes, capacity=cap, device=local_rank, timing=True)

    d_rating = torch.from_numpy(rating).cuda()
    d_cons = torch.from_numpy(cons.view(np.int32)).cuda()
    eng = Engine(cfg)

    def step():
        eng.reset()
        eng.enqueue_device(d_rating, d_cons)
        m = eng.tick(0)
        return m, dict(eng.last_enqueue_stats)

    def fence():
        torch.cuda.synchronize()
        if dist_on:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    walk_ms, bucket_ms, copy_ms = [], [], []
    last = None
    for _ in range(args.steps):
        last, est = step()
        walk_ms.append(last.stats["walk_ms"])
        bucket_ms.append(est["bucket_ms"])
        copy_ms.append(last.stats["copy_ms"])
    fence()
    elapsed = time.perf_counter() - t0

    matched = float(last.stats["players_matched"]) * args.steps
    pairs = float(last.stats["pairs"])
    if dist_on:
        t = torch.tensor([elapsed], device="cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        s = torch.tensor([matched], device="cuda", dtype=torch.float64)
        dist.all_reduce(s, op=dist.ReduceOp.SUM)
        matched = float(s.item())

    if rank == 0:
        k_ms = float(np.mean(walk_ms))
        traffic = None
        try:
            tpath = args.traffic_json or os.path.join(
                ROOT, "profiles", "traffic_latest.json" if args.mode == "1v1" else "traffic_latest_%s.json" % args.mode)
            with open(tpath) as f:
                tj = json.load(f)
            if tj.get("workload_players") == n and tj.get("mode") == args.mode:
                traffic = tj.get("walk_hbm_bytes_per_tick")
        except Exception:
            pass
        line = {
            "metric": "matched players/sec over 1M-player pool",
            "value": matched / elapsed,
            "unit": "matched players/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "int32",
            "data": "synthetic",
            "config": {"workload": workload, "rating_groups": 7, "pool_per_gpu": n,
                       "sharding": "one engine + own pool per GPU; no data-path collective",
                       "step": "reset + enqueue(device-resident) + search to quiescence + match list D2H"},
            "matched_fraction": float(last.stats["players_matched"]) / n,
            "pair_evals_per_s": pairs / (k_ms * 1e-3) if k_ms > 0 else None,
            "pairs_per_step": pairs,
            "passes_max": last.stats["passes_max"],
            "kernel_ms": {"walk": k_ms, "bucket(count+scan+scatter)": float(np.mean(bucket_ms)),
                          "d2h+bookkeeping": float(np.mean(copy_ms))},
            "roofline": roofline_block(args.mode, pairs, bytes_per_pair, k_ms, elapsed / args.steps * 1e3, n,
                                       last.stats["passes_max"], traffic),
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(cfg, rating, cons, args.mode, budget_s=args.cpu_baseline_seconds)
        if world == 1 and not args.no_stream and args.mode == "1v1":
            scfg = make_config(modes, capacity=1 << 20, device=local_rank, timing=False)
            line["latency"] = stream_latency(lambda: Engine(scfg), args.stream_qps, args.stream_seconds,
                                             args.stream_tick_ms, "1v1 +-%d, region filter" % args.window)
            # BASELINE cfg-5 on one GPU: the same stream with 70 % 1v1 / 30 % 5v5 (roles as cfg-3)
            mcfg = make_config([mode_1v1(window=args.window, region_filter=True), mode_team(5, 2, 50, (1, 1, 1, 1, 1))],
                               capacity=1 << 20, device=local_rank, timing=False)
            line["latency_mixed"] = stream_latency(lambda: Engine(mcfg), args.stream_qps, args.stream_seconds,
                                                   args.stream_tick_ms,
                                                   "70 %% 1v1 +-%d region filter / 30 %% 5v5 +-50 five roles" % args.window,
                                                   mode_weights=(70, 30), role_weights=ROLE_WEIGHTS_5V5)
        if world == 1 and args.concurrent_pools > 1:
            ccfg = make_config(modes, capacity=cap, device=local_rank, timing=False)

            def pool_inputs(k):
                kw = {"role_weights": ROLE_WEIGHTS_5V5} if args.mode == "5v5" else {}
                r, c = make_pool(n, seed=101 + k, dist=args.dist, **kw)
                return torch.from_numpy(r).cuda(), torch.from_numpy(c.view(np.int32)).cuda()

            line["concurrent_pools"] = concurrent_pools(lambda: Engine(ccfg), args.concurrent_pools, args.steps,
                                                        pool_inputs)
        print(json.dumps(line), flush=True)
    eng.close()
    if dist_on:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
