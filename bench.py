#!/usr/bin/env python
"""bench.py — matched players/sec over a synthetic pool (BASELINE.json metric), on N MI355X.

One "step" = one pass of the hot path over one batch of synthetic input: reset the engine,
enqueue the rank's share of the pool, ALREADY RESIDENT IN HBM (bucketing kernels), run the
search to quiescence (walk kernels), and bring the match list back to the host (the service
needs it to publish lobbies).

N = 1: BASELINE.json configs[1] — "1v1, 1M players, +-25 rating + region filter on 1 MI355X",
seeded synthetic pool (uniform integer ratings on [0, 5000], 8 regions, arrival order = index).

N > 1: BASELINE.json configs[3] — ONE 10M-player 1v1 pool sharded across the N GPUs.  Every
rank generates the same seeded pool and keeps the chains = (game mode, rating group) that
`sharding.ChainSharding` gives it (reference lib/application.ex:26-40: one queue, one lobby
table and one worker per rating group; lib/models/lobby_state.ex:74-83: the stored lobby is
selected by game mode).  Chains never interact, so there is NO data-path collective: RCCL is
used for the barriers and for gathering the result scalars / digests.  `value` = matched players
of the whole pool / the slowest rank's time (strong scaling: the pool is fixed as N grows); it
is bounded by the heaviest chain (the first rating group holds 30 % of a uniform pool), and with
7 chains on 8 ranks one rank idles — reported as such (`config.sharding`).  The union of the
ranks' emission lists is checked against the oracle's digest of the same pool
(tests/golden/shared_pool_digests.json, tools/make_shared_pool_digests.py) — `exactness`.
Secondary keys, never `value`: `weak_scaling` (every rank its own 1M pool) and `latency_mixed`
(BASELINE configs[4]: the 100k players/s 70/30 stream, chains sharded over the ranks).

Prints ONE JSON line (rank 0).  `roofline` is for the walk (the search proper: for a 1v1
mode the pair path's kernel sequence kp_nx_init, kp_round / kp_rounds (one launch per pass of the cursor / a batch
of passes per launch), kp_late, kp_finish — DESIGN.md §4; for team modes kt_build, kt_f | kt_f2 | kt_chase or one
kt_fc launch per pass, kt_late):
achieved = algorithmic bytes / HIP-event time of that sequence on the engine's stream,
algorithmic bytes = pair evaluations of the reference algorithm (the oracle's count, which
the engine reproduces bit-exactly) x 8 B (rating + cons of the candidate).
The engine does NOT touch every such pair — it keeps next[] pointers instead of rescanning —
so `traffic` (HBM bytes from rocprofv3 PMC passes, profiles/) is unrelated to the algorithmic
bytes; the walk is bound by the latency of several hundred dependent passes, not by HBM
(DESIGN.md §5): `roofline.latency_floor_ms` = passes x one dependent kernel boundary (measured
in this run) is the ceiling the design can be held to, `frac_of_latency_ceiling` how close it is.
`cpu_baseline` = the oracle (oracle/mode_r.c, a C port of the reference's sequential search,
in-memory — an upper bound on what the BEAM service could do) timed on this box's host cores.
"""
import argparse
import hashlib
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
CSRC = os.path.join(ROOT, "microservice_matchmaking_amd", "csrc")
DIGESTS = os.path.join(ROOT, "tests", "golden", "shared_pool_digests.json")


def kernel_source_hash():
    """Hash of the kernel sources a traffic / profile file was measured on."""
    h = hashlib.blake2b(digest_size=8)
    for name in sorted(os.listdir(CSRC)):
        if name.endswith((".hip", ".inc", ".h")):
            with open(os.path.join(CSRC, name), "rb") as f:
                h.update(name.encode() + b"\0" + f.read())
    return h.hexdigest()


def load_traffic(path, players, mode, detail=None):
    """PMC traffic of one tick from profiles/ — only if it was measured on THESE kernel sources
    and this workload; otherwise (None, why).  `detail` (a dict) receives who owns the bytes: the three kernels with
    the most traffic per tick (FETCH_SIZE doubled + WRITE_SIZE, as the total) and, when the file holds it, the predicate
    tests the kernels physically performed (MM_PAIR_TUNE=0x2000 counter, a run of its own)."""
    try:
        with open(path) as f:
            tj = json.load(f)
    except Exception:
        return None, "no PMC file %s" % os.path.basename(path)
    if tj.get("workload_players") != players or tj.get("mode") != mode:
        return None, "PMC file is for another workload"
    if tj.get("source_hash") != kernel_source_hash():
        return None, "PMC file %s was measured on other kernel sources (hash %s, now %s): re-run tools/make_traffic.py" % (
            os.path.basename(path), tj.get("source_hash"), kernel_source_hash())
    if detail is not None:
        per = tj.get("per_kernel_kb_per_tick") or {}
        rows = sorted(((2.0 * v.get("fetch_raw", 0.0) + v.get("write_raw", 0.0)) * 1024.0, k, v) for k, v in per.items())
        detail["traffic_by_kernel"] = [{"kernel": k, "bytes_per_tick": b, "dispatches_per_tick": v.get("dispatches_per_tick")}
                                       for b, k, v in reversed(rows[-3:])]
        # round 6: every walk kernel's bytes AND time per tick (the rocprofv3 summaries the file was made from), heaviest first
        tm = tj.get("per_kernel_time") or {}
        detail["by_kernel_profiled"] = [
            {"kernel": k, "hbm_bytes_per_tick": b, "launches_per_tick": v.get("dispatches_per_tick"),
             "us_per_tick": (tm.get(k) or {}).get("us_per_tick"), "avg_us_per_launch": (tm.get(k) or {}).get("avg_us"),
             "gbs": (b / ((tm.get(k) or {}).get("us_per_tick") * 1e-6) / 1e9) if (tm.get(k) or {}).get("us_per_tick") else None}
            for b, k, v in reversed(rows)]
    return tj.get("walk_hbm_bytes_per_tick"), None


# The serial critical path of a pair tick, from primitives measured on this part (cycles at 2.4 GHz; the files under profiles/):
#   a pass inside kp_rounds   = the chain's flag barrier + its route hops at the L2-hit price
#   a pass as a kp_round launch = a kernel boundary (measured in this run) + its hops at the price of a load from memory
#   kp_late                   = one dependent LDS step per lobby + one per pass
# Everything else a pass does today (the tile work: apply, repair, tables) is parallel work that COULD be hidden; this is the
# floor of the design as built, what `frac_of_critical_path` holds the walk to.
CRIT = {"barrier_us": 4277 / 2400.0,          # profiles/r04_pair_phase_timers.txt, g0 kp_rounds: "barrier 4277" cycles a pass
        "hop_l2_us": 281 / 2400.0,            # a route hop whose line sits in the chain's L2 (tw_hops)
        "hop_mem_us": 470 / 2400.0,           # profiles/r04_ubench_xwg_hop.txt: a dependent load of a line that was only stored
        "late_step_us": 100 / 2400.0,         # the blind chase of kp_late, cycles per lobby
        "late_pass_us": 128 / 2400.0,         # one dependent LDS round trip (the pass's careful last step)
        "boundary_us_default": 2.4}           # kp_round to kp_round (profiles/r02_kernel_passes_1m_1v1.txt) when not measured here


def measured_primitives(path):
    """The pair chain's serial primitives as THIS run's tick timed them (mm_path_stats, round 6): the shader clock from kp_late's
    chase (cycles over 100 MHz ticks), the flag barrier and the L2-hit hop from tile 1's walker of the critical chain.  Falls
    back, field by field, to the round-4 constants of CRIT (a library without the counters, a tick without kp_rounds)."""
    prim = dict(CRIT)
    src = {k: "constant (profiles/r04_*)" for k in ("barrier_us", "hop_l2_us")}
    mhz = 2400.0
    if path and path.get("clk_cycles") and path.get("clk_wall_ticks"):
        mhz = 100.0 * path["clk_cycles"] / path["clk_wall_ticks"]
        src["clock_mhz"] = "kp_late's chase: clk_cycles / clk_wall_ticks x 100 MHz"
    else:
        src["clock_mhz"] = "constant 2400"
    if path and path.get("crit_timed_passes") and path.get("crit_barrier_cycles"):
        prim["barrier_us"] = path["crit_barrier_cycles"] / path["crit_timed_passes"] / mhz
        src["barrier_us"] = "measured in this run: crit_barrier_cycles / crit_timed_passes / clock"
    if path and path.get("crit_timed_hops") and path.get("crit_hop_cycles"):
        prim["hop_l2_us"] = path["crit_hop_cycles"] / path["crit_timed_hops"] / mhz
        src["hop_l2_us"] = "measured in this run: crit_hop_cycles / crit_timed_hops / clock"
    prim["clock_mhz"] = mhz
    return prim, src


def critical_path_ms(path, boundary_us=None, prim=None):
    """(ms, parts) of the pair path's serial critical path for the tick `path` (mm_path_stats_get) describes, or (None, None)."""
    if not path or not (path.get("paths", 0) & 2) or not path.get("crit_passes") or path.get("crit_group") == 0xFFFFFFFF:
        return None, None
    prim = prim or CRIT
    b = boundary_us if boundary_us else CRIT["boundary_us_default"]
    rp, rh = path["crit_rounds_passes"], path["crit_rounds_hops"]
    hops_per_pass = (rh / rp) if rp else 0.0
    parts = {"kp_rounds": (rp * prim["barrier_us"] + rh * prim["hop_l2_us"]) * 1e-3,
             "kp_round": path["crit_round_passes"] * (b + hops_per_pass * prim["hop_mem_us"]) * 1e-3,
             "kp_late": (path["crit_late_lobbies"] * prim["late_step_us"] + path["crit_late_passes"] * prim["late_pass_us"]) * 1e-3}
    return sum(parts.values()), parts


# The team path's serial chain (round 6): what the critical chain's CHASER must do one thing after the other.
#   a pass as kt_f | kt_f2 | kt_chase  = three kernel boundaries + one dependent (F, F o F) load per TWO lobbies, pulled through the
#                                         chaser's L2 (169 cycles: profiles/r04_ubench_xwg_hop.txt "pulled")
#   a pass as ONE kt_fc launch         = one boundary + one dependent F load per lobby from memory (470 cycles, same file)
#   a pass inside kt_late              = no boundary; one trip to memory per lobby (the anchor's record)
#   a look-up the chaser does itself   = four dependent trips to memory (where the roles' stretches begin, the stretches, the
#                                         members' records, the bitmap behind the last member) at 982 cycles a trip on an idle
#                                         device (profiles/r03_ubench_trip_latency.txt), three inside kt_late (the bitmap is in LDS)
# kt_f's chunks, the emitters and the seats are parallel work beside it.
TEAM_CRIT = {"hop_pulled_us": 169 / 2400.0, "hop_mem_us": 470 / 2400.0, "trip_us": 982 / 2400.0,
             "lookup_trips": 4, "late_lookup_trips": 3}


def team_critical_path_ms(path, boundary_us=None):
    """(ms, parts) of the team path's serial critical path from mm_path_stats' crit_team_* counts, or (None, None)."""
    if not path or not (path.get("paths", 0) & 4) or not path.get("crit_team_passes") or path.get("crit_team_group") in (None, 0xFFFFFFFF):
        return None, None
    b = boundary_us if boundary_us else CRIT["boundary_us_default"]
    T = TEAM_CRIT
    parts = {"kt_f|kt_f2|kt_chase": (path["crit_team_f_passes"] * 3 * b + 0.5 * path["crit_team_f_lobbies"] * T["hop_pulled_us"]) * 1e-3,
             "kt_fc": (path["crit_team_fc_passes"] * b + path["crit_team_fc_lobbies"] * T["hop_mem_us"]) * 1e-3,
             "kt_late": path["crit_team_late_lobbies"] * T["trip_us"] * 1e-3,
             "look-ups": (path["crit_team_lookups"] * T["lookup_trips"] + path["crit_team_late_lookups"] * T["late_lookup_trips"]) * T["trip_us"] * 1e-3}
    return sum(parts.values()), parts


def roofline_block(mode, pairs, bytes_per_pair, walk_ms, step_ms, players, passes_max, traffic,
                   boundary_us=None, traffic_note=None, traffic_detail=None, path=None, probe=None):
    """The `roofline` object of the bench line (pure arithmetic; tests/test_bench_line.py).
    SURVEY.md §8(d): achieved = algorithmic bytes / walk time, with its mandatory companions —
    (i) physical HBM GB/s (PMC traffic / walk time), (ii) the compulsory bytes of a tick
    (every player read once, 12 B, and written out once, 8 B) and how close the whole step is
    to streaming just those, (iii) the tile width of the dominant kernel — and the latency
    ceiling of a chain of dependent passes (passes x one kernel boundary)."""
    walk_name = ("pair walk: kp_nx_init + kp_rounds (a batch of passes per launch, tiles LDS resident; kp_round, one launch "
                 "per pass, while a chain does not fit one XCD and as its fallback) + kp_late + kp_finish"
                 if mode == "1v1" else
                 "team walk: kt_build + per pass ONE launch kt_fc (kt_f, the chase and the emitters side by side; the first 32 "
                 "lobby-rich passes kt_f + kt_f2 + kt_chase with its emitters) + kt_late (the last passes in one launch)")
    achieved = pairs * bytes_per_pair / (walk_ms * 1e-3) / 1e9 if walk_ms > 0 else 0.0
    compulsory = float(players) * (12 + 8)
    floor_ms = passes_max * boundary_us * 1e-3 if boundary_us else None
    out = {
        "bound": "hbm", "kernel": walk_name, "achieved": achieved, "peak": HBM_PEAK_GBS,
        "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
        "algorithmic_bytes_per_launch": pairs * bytes_per_pair,
        "launch": "one tick = one walk to quiescence (HIP events around the kernel sequence)",
        "frac_of_measured_copy_peak": achieved / HBM_COPY_GBS,
        "physical_gbs": (traffic / (walk_ms * 1e-3) / 1e9) if (traffic and walk_ms > 0) else None,
        "compulsory_bytes_per_tick": compulsory,
        "compulsory_frac": (compulsory / (step_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if step_ms > 0 else None,
        "tile_positions": 8192 if mode == "1v1" else 512,
        "boundary_us": boundary_us,
        "latency_floor_ms": floor_ms,
        "frac_of_latency_ceiling": (floor_ms / walk_ms) if (floor_ms and walk_ms > 0) else None,
        "us_per_pass": (walk_ms * 1e3 / passes_max) if passes_max else None,
        "critical_path_ms": None, "frac_of_critical_path": None,
        "note": ("Mode R is a chain of dependent first-fit steps (%d passes of the cursor for "
                 "the longest rating group); the engine replaces the per-pair rescans by "
                 "%s, so it is bound by pass latency (kernel boundaries + LDS/VALU issue), "
                 "not by HBM; latency_floor_ms = passes x one dependent kernel boundary is the "
                 "ceiling of a one-launch-per-pass design; see DESIGN.md") % (
                    passes_max,
                    "next[] pointers repaired incrementally" if mode == "1v1" else
                    "per-pass F pointers (who the cursor picks after each player's lobby) "
                    "computed for every queued player at once"),
    }
    prim, prim_src = measured_primitives(path)
    cp, cparts = critical_path_ms(path, boundary_us, prim)
    tcp, tparts = team_critical_path_ms(path, boundary_us)
    if tcp is not None:
        out["critical_path_ms"] = tcp
        out["frac_of_critical_path"] = (tcp / walk_ms) if walk_ms > 0 else None
        out["critical_path_model"] = {
            "parts_ms": tparts, "primitives_us": dict(TEAM_CRIT, boundary_us=boundary_us or CRIT["boundary_us_default"]),
            "passes": {"kt_f|kt_f2|kt_chase": path["crit_team_f_passes"], "kt_fc": path["crit_team_fc_passes"], "kt_late": path["crit_team_late_passes"]},
            "lobbies": {"by F o F hops": path["crit_team_f_lobbies"], "by F hops": path["crit_team_fc_lobbies"], "inside kt_late": path["crit_team_late_lobbies"]},
            "lookups": {"pass kernels": path["crit_team_lookups"], "kt_late": path["crit_team_late_lookups"]},
            "what": "the chain with the most passes (rating group %d), its chaser alone: kernel boundaries, one dependent load per lobby "
                    "(per two with F o F), four dependent trips to memory per look-up it does itself (the stored lobby's fill, the lobby a "
                    "pass ends on); kt_f's chunks, the emitters and the seats are parallel work beside it" % path["crit_team_group"]}
    if cp is not None:
        # latency_floor_ms above is the floor of a one-launch-per-pass design and is kept for continuity with rounds 1-4;
        # the path that runs now has no kernel boundary in most passes: this is the ceiling that applies to it
        out["critical_path_ms"] = cp
        out["frac_of_critical_path"] = (cp / walk_ms) if walk_ms > 0 else None
        out["critical_path_model"] = {
            "parts_ms": cparts, "primitives_us": {k: v for k, v in prim.items() if k != "boundary_us_default"},
            "primitives_source": prim_src,
            "passes": {"kp_rounds": path["crit_rounds_passes"], "kp_round": path["crit_round_passes"], "kp_late": path["crit_late_passes"]},
            "hops_in_kp_rounds": path["crit_rounds_hops"], "lobbies_in_kp_late": path["crit_late_lobbies"],
            "what": "the chain with the most passes (rating group %d): barrier + route hops per pass inside kp_rounds, kernel boundary "
                    "+ hops from memory per kp_round launch, one dependent LDS step per lobby and pass inside kp_late" % path["crit_group"]}
    if traffic is None and traffic_note:
        out["traffic_note"] = traffic_note
    if traffic is not None:
        out["traffic_over_algorithmic"] = traffic / (pairs * bytes_per_pair) if pairs else None
    out["traffic_by_kernel"] = (traffic_detail or {}).get("traffic_by_kernel")
    # round 6: the walk's kernels one by one.  Time and HBM bytes per tick are the profile's (the same
    # rocprofv3 summaries `traffic` comes from, bound to the kernel sources by hash); kp_nx_init — north_star's kernel: coalesced
    # SoA keys, an LDS-staged candidate window, a wave-ballot arg-min per anchor — is also timed LIVE in this run (HIP events on the
    # engine's stream) and priced by SURVEY.md 8(d)'s convention: candidates it physically tested x 8 B / its duration.  That figure
    # is LDS traffic, not HBM traffic: it is labelled as such and is never `frac`.
    by = list((traffic_detail or {}).get("by_kernel_profiled") or [])
    tests = (probe or {}).get("tests_all")
    out["predicate_tests_physical"] = tests
    if mode == "1v1" and path and path.get("pair_nx_init_ns"):
        us = path["pair_nx_init_ns"] / 1e3
        nx = {"kernel": "kp_nx_init", "duration_us_live": us, "tested_candidates": (probe or {}).get("tests_nx_init"),
              "label": "LDS-staged, SURVEY 8(d) convention (8 B per tested candidate); LDS bytes, not HBM bytes: never `frac`"}
        if nx["tested_candidates"]:
            gbs = nx["tested_candidates"] * 8.0 / (us * 1e-6) / 1e9
            nx.update({"equiv_gbs": gbs, "equiv_frac_of_hbm_peak": gbs / HBM_PEAK_GBS,
                       "tests_per_s": nx["tested_candidates"] / (us * 1e-6)})
        for row in by:
            if row["kernel"].startswith("kp_nx_init"):
                row.update(nx)
                break
        else:
            by.append(nx)
    out["by_kernel"] = by or None
    if tests is not None:
        out["predicate_tests_note"] = ("all pair kernels of one tick (next[] build, repairs, head scans, the LDS-resident walk), counted by a "
                                       "second engine created with mm_tuning.pair_tune bit 13 (atomics: not the timed engine); "
                                       "%.2f x the reference algorithm's pair evaluations" % (tests / pairs if pairs else 0.0))
    return out


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=180, help="default: a timed region of a good 2 s at N=1 (12 ms per step)")
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--players", type=int, default=None,
                    help="pool size; default 1,000,000 at N=1 (cfg-2) and 10,000,000 at N>1 (cfg-4, one shared pool)")
    ap.add_argument("--window", type=int, default=25)
    ap.add_argument("--dist", default="uniform", choices=["uniform", "normal"])
    ap.add_argument("--mode", default="1v1", choices=["1v1", "5v5"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-seconds", type=float, default=20.0, help="CPU time budget of the cpu_baseline leg")
    ap.add_argument("--no-stream", action="store_true", help="skip the streaming-latency legs")
    ap.add_argument("--no-saturation", action="store_true", help="skip the latency_saturation leg (1v1 stream at rising rates)")
    ap.add_argument("--saturation-qps", default="100000,1000000,5000000,20000000", help="offered rates of the latency_saturation leg")
    ap.add_argument("--saturation-seconds", type=float, default=1.0)
    ap.add_argument("--no-boundary", action="store_true",
                    help="skip the kernel-boundary micro-measurement (torch add kernels; tools/collect_profiles.sh "
                         "passes this so that the profiled command holds the engine's kernels only)")
    ap.add_argument("--no-pcie", action="store_true", help="skip the pcie_inclusive leg (the pool handed over in host buffers)")
    ap.add_argument("--no-cfg3", action="store_true", help="skip the cfg3 leg (BASELINE configs[2] beside the main line)")
    ap.add_argument("--cfg3-steps", type=int, default=12)
    ap.add_argument("--no-prediction", action="store_true",
                    help="skip sharding_prediction (every rank's share of the 10M pool run alone on this GPU)")
    ap.add_argument("--no-secondary", action="store_true",
                    help="skip the secondary legs (weak_scaling at N>1, shared_pool_n1 and concurrent_pools at N=1)")
    ap.add_argument("--concurrent-pools", type=int, default=2,
                    help="secondary leg (N=1 only): this many engines, each with its own pool of --players, "
                         "ticking concurrently on their own streams; reported beside the main line, never as `value`")
    ap.add_argument("--shared-players", type=int, default=10_000_000, help="pool of the shared_pool_n1 leg (cfg-4 on one GPU)")
    ap.add_argument("--weak-players", type=int, default=1_000_000, help="pool per rank of the weak_scaling leg (N>1)")
    ap.add_argument("--stream-qps", type=int, default=100_000)
    ap.add_argument("--stream-seconds", type=float, default=3.0)
    ap.add_argument("--stream-tick-ms", type=float, default=10.0)
    ap.add_argument("--force-dist", action="store_true",
                    help="N = 1 with the process group all the same: RCCL (`nccl`) is initialised on the one GPU and the barrier, the "
                         "all-reduce on the device and the gathers of the N > 1 branch run for real, on one rank — what a 1-GPU box can "
                         "execute of that branch with the backend the driver's 8-GPU run uses")
    ap.add_argument("--no-probe", action="store_true",
                    help="skip the extra untimed tick that counts the physical predicate tests (tools/collect_profiles.sh: "
                         "a profile divided by its ticks must hold the timed engine's ticks only)")
    ap.add_argument("--same-device", action="store_true",
                    help="N > 1 ranks that all use GPU 0 and meet over gloo instead of RCCL (two RCCL ranks cannot share a device): "
                         "what a 1-GPU box can run of the N > 1 branch — real HIP engines in N processes, the sharding, the "
                         "gathers, the one JSON line; the line says same_device and its value is NOT a scaling number")
    ap.add_argument("--traffic-json", default=None,
                    help="optional PMC summary written by tools/make_traffic.py (default: profiles/traffic_latest*.json)")
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


def measure_boundary_us(torch, n=400):
    """One dependent kernel boundary on this GPU: a hipGraph of n trivial dependent launches,
    HIP-event time / n (MI355X_MICROARCH.md `boundary`: ~1.45 us).  The walk is one launch per
    pass of the cursor, so passes x this is the floor of the design."""
    try:
        x = torch.zeros(256, device="cuda", dtype=torch.int32)
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            for _ in range(3):
                x.add_(1)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=s):
                for _ in range(n):
                    x.add_(1)
            g.replay()
            torch.cuda.synchronize()
            best = None
            for _ in range(5):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(s)
                g.replay()
                e1.record(s)
                torch.cuda.synchronize()
                t = e0.elapsed_time(e1) * 1e3 / n
                best = t if best is None or t < best else best
        return float(best)
    except Exception:
        return None


def concurrent_pools(make_engine, pools, steps, make_inputs, digest_of=None, expected=None):
    """Secondary leg: the walk is bound by the latency of its passes and keeps well under half of the CUs
    busy, so independent pools (other regions / shards of a service) on engines of their own
    (own stream, own device memory: include/mm_engine.h) overlap.  `pools` host threads, one
    engine each, `steps` steps each (the ctypes calls release the GIL); matched players of all
    pools over the wall time of the slowest."""
    import threading
    engines = [make_engine() for _ in range(pools)]
    inputs = [make_inputs(k) for k in range(pools)]
    matched = [0] * pools
    lasts = [None] * pools
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
                lasts[k] = eng.tick(0)
                matched[k] += int(lasts[k].stats["players_matched"])
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
    # every pool's LAST tick, walked while the other engines were ticking, against the oracle's digest of that seeded pool
    exact = None
    if digest_of is not None and expected is not None:
        exact = []
        for k in range(pools):
            want = expected(k)
            exact.append((digest_of(lasts[k]) == want) if want else None)
    return {"pools": pools, "steps": steps, "value": sum(matched) / elapsed, "unit": "matched players/s",
            "ms_per_step_per_pool": elapsed / steps * 1e3, "exact_per_pool": exact,
            "ok": (all(x for x in exact) if exact and all(x is not None for x in exact) else None),
            "note": "k independent pools on one GPU, one engine and stream each; not the headline workload "
                    "(measured sweet spot: 2 pools, profiles/r02_concurrent_pools_*.json)"}


def tick_cost_windows(cost, tick_ms, window_s=10.0):
    """Per-tick host+device cost of a stream, summarised per window of the stream's own clock."""
    cost = np.asarray(cost, dtype=np.float64) * 1e3
    per = max(1, int(round(window_s * 1000.0 / tick_ms)))
    out = []
    for k in range(0, len(cost), per):
        c = cost[k:k + per]
        out.append({"from_s": k * tick_ms * 1e-3, "p50": float(np.percentile(c, 50)), "p99": float(np.percentile(c, 99)),
                    "max": float(c.max())})
    return out


def stream_leg(search, dist, rank, world, qps, seconds, tick_ms, label, key_label, seed=77, mode_weights=None,
               role_weights=None):
    """Second half of BASELINE.json's metric: match latency at a fixed enqueue rate
    (microservice_matchmaking_amd/stream.py).  Every rank sees the whole stream and keeps its
    chains; rank 0 gathers the latencies of all ranks.  The stream is tick-count driven, so what it
    emits is deterministic: the union digest, the matched players and the backlog are checked against
    the oracle's run of the same schedule (tests/golden/shared_pool_digests.json)."""
    from microservice_matchmaking_amd.sharding import union_digest
    from microservice_matchmaking_amd.stream import latency_summary, run_stream, stream_batch, stream_schedule
    n_modes = int(search.cfg.n_modes)
    r0, c0 = stream_batch(2000, seed, mode_weights, role_weights)   # warm the kernels up outside the clock
    search.enqueue(r0, c0)
    for md in range(n_modes):
        search.tick(md)
    search.engine.reset()
    sched = stream_schedule(qps, seconds, tick_ms, seed)
    if dist is not None:
        dist.barrier()
    res = run_stream(search, sched, mode_weights=mode_weights, role_weights=role_weights, realtime=True)
    mine = {c: d for c, d in res["digests"].items() if search.sharding.chain_owner[c] == rank}
    part = {"real": res["real"], "floor": res["floor"], "matched": res["matched"], "elapsed": res["elapsed"],
            "tick_cost": res["tick_cost"], "depth": [d.tolist() for d in res["depth"]], "digests": mine,
            "full_at_s": res.get("full_at_s")}
    parts = [part]
    if dist is not None:
        parts = [None] * world
        dist.all_gather_object(parts, part)
    if rank != 0:
        return None
    cat = lambda key, md: np.concatenate([p[key][md] for p in parts])
    per_mode = []
    for md in range(n_modes):
        s = latency_summary(cat("real", md), cat("floor", md))
        s["backlog_players"] = int(sum(sum(p["depth"][md]) for p in parts))
        per_mode.append(s)
    real = np.concatenate([cat("real", md) for md in range(n_modes)])
    floor = np.concatenate([cat("floor", md) for md in range(n_modes)])
    elapsed = max(p["elapsed"] for p in parts)
    cost = np.max(np.stack([np.asarray(p["tick_cost"]) for p in parts]), axis=0) if len({len(p["tick_cost"]) for p in parts}) == 1 \
        else np.concatenate([p["tick_cost"] for p in parts])
    digests = {}
    for p in parts:
        digests.update(p["digests"])
    matched = int(sum(p["matched"] for p in parts))
    got = union_digest(digests)
    want = expected_digest(stream_key(key_label, qps, seconds, tick_ms, seed))
    out = {"enqueue_qps": qps, "tick_ms": tick_ms, "seconds": seconds, "mode": label, "ranks": world,
           "capacity": int(search.cfg.capacity)}
    out.update(latency_summary(real, floor))
    backlog = [int(s["backlog_players"]) for s in per_mode]
    out.update({
        "matched_players_per_s": matched / elapsed,
        "tick_cost_ms_mean": float(np.mean(cost) * 1e3), "tick_cost_ms_p50": float(np.percentile(cost, 50) * 1e3),
        "tick_cost_ms_p99": float(np.percentile(cost, 99) * 1e3), "tick_cost_ms_max": float(np.max(cost) * 1e3),
        "tick_cost_ms_by_10s": tick_cost_windows(cost, tick_ms),
        "backlog_players": int(sum(backlog)),
        "kept_up": bool(elapsed < seconds * 1.05),
        "elapsed_s": elapsed,
        "capacity_exhausted_at_s": next((p["full_at_s"] for p in parts if p["full_at_s"] is not None), None),
        "emission_digest": got,
        "oracle_digest": want["digest"] if want else None,
        "ok": (got == want["digest"] and matched == want["matched"]
               and backlog == [int(sum(b)) for b in want["backlog"]]) if want else None,
        "oracle_matched": want["matched"] if want else None,
        "note": "floor_* = end of the tick period in which the player was matched minus its arrival: the wait for "
                "fitting partners to ARRIVE (reference behaviour, docs/MATCH_CHECK.md section 4: a chain whose "
                "anchor nobody fits waits for arrivals), what an engine with a free tick would give; "
                "engine_added = real - floor; ok = emission digest, matched players and backlog per mode equal the "
                "CPU oracle's run of the same tick schedule (committed digest; None: no digest for this schedule)"})
    if n_modes > 1:
        out["per_mode"] = per_mode
    return out


def workload_key(mode, players, window, dist_name, seed=1):
    return "%s/%d/w%d/%s/seed%d" % (mode, players, window, dist_name, seed)


def stream_key(label, qps, seconds, tick_ms, seed=77):
    """Key of a stream leg in tests/golden/shared_pool_digests.json (label: "1v1" | "mixed")."""
    return "stream/%s/qps%d/s%g/tick%g/seed%d" % (label, qps, seconds, tick_ms, seed)


def saturation_leg(make_search, window, tick_ms=10.0, seconds=1.0, qps_list=(100_000, 1_000_000, 5_000_000, 20_000_000), seed=77):
    """The latency axis with the ENGINE as the bottleneck: a 1v1 stream at rising offered rates, in `tick_ms` ticks,
    until it no longer keeps up — a tick (H2D of the period's arrivals, bucketing, the walk to quiescence, the match list
    back) has to fit its period.  The arrivals of a leg are made before its clock starts (numpy needs longer than 10 ms
    for 200 000 players).  `sustained` = the highest rate with every tick's cost p99 inside the period and the stream over
    in time; the match latency there is the arrival-limited floor plus that tick cost.  The leg at the rate for which
    tests/golden/shared_pool_digests.json holds the oracle's digest of the same schedule is checked against it."""
    from microservice_matchmaking_amd.sharding import union_digest
    from microservice_matchmaking_amd.stream import latency_summary, run_stream, stream_batch, stream_schedule
    legs, sustained = [], None
    for qps in qps_list:
        search = make_search(stream_capacity(qps, seconds))
        try:
            r0, c0 = stream_batch(2000, seed)
            search.enqueue(r0, c0)
            search.tick(0)
            search.engine.reset()
            sched = stream_schedule(qps, seconds, tick_ms, seed)
            batches = [stream_batch(n, sd) for (_, _, n, sd, _) in sched]
            res = run_stream(search, sched, realtime=True, batches=batches)
        finally:
            search.close()
        cost = np.asarray(res["tick_cost"]) * 1e3
        lat = latency_summary(np.concatenate([x for x in res["real"]]), np.concatenate([x for x in res["floor"]]))
        key = stream_key("1v1", qps, seconds, tick_ms, seed)
        want = expected_digest(key)
        kept = bool(res["elapsed"] < seconds * 1.05 and res["full_at_s"] is None and len(cost) and np.percentile(cost, 99) < tick_ms)
        leg = {"enqueue_qps": qps, "players_per_tick": int(round(qps * tick_ms * 1e-3)), "kept_up": kept,
               "tick_cost_ms_p50": float(np.percentile(cost, 50)) if len(cost) else None,
               "tick_cost_ms_p99": float(np.percentile(cost, 99)) if len(cost) else None,
               "tick_cost_ms_max": float(cost.max()) if len(cost) else None,
               "matched_players_per_s": res["matched"] / res["elapsed"], "elapsed_s": res["elapsed"],
               "p50_ms": lat["p50_ms"], "p99_ms": lat["p99_ms"], "floor_p99_ms": lat["floor_p99_ms"],
               "engine_added_p99_ms": lat.get("engine_added_p99_ms"),
               "backlog_players": int(sum(int(d.sum()) for d in res["depth"])),
               "capacity_exhausted_at_s": res["full_at_s"]}
        if want is not None:
            got = union_digest(res["digests"])
            leg["exactness"] = {"key": key, "emission_digest": got, "oracle_digest": want["digest"],
                                "ok": bool(got == want["digest"] and res["matched"] == want["matched"])}
        legs.append(leg)
        if kept:
            sustained = leg
        else:
            break
    return {"mode": "1v1 +-%d, region filter" % window, "tick_ms": tick_ms, "seconds_per_leg": seconds, "legs": legs,
            "highest_sustained_qps": sustained["enqueue_qps"] if sustained else None,
            "tick_cost_ms_p50_there": sustained["tick_cost_ms_p50"] if sustained else None,
            "tick_cost_ms_p99_there": sustained["tick_cost_ms_p99"] if sustained else None,
            "p99_ms_there": sustained["p99_ms"] if sustained else None,
            "ok": (all(l["exactness"]["ok"] for l in legs if "exactness" in l) if any("exactness" in l for l in legs) else None),
            "note": "offered rate raised until a tick no longer fits its period: what the ENGINE sustains, beside the "
                    "arrival-limited floors of `latency` / `latency_mixed` (BASELINE cfg-5's 100k players/s is 1 % of it)"}


def stream_capacity(qps, seconds):
    """Slot ring of a stream engine.  A stream's pool is what has not been matched yet; cfg-5's 5v5 half has no
    steady state (the role weights cap its match rate at half of its arrivals), so the pool
    grows by about 15 % of the arrivals: sized at 30 % of everything that arrives, at least 1M slots."""
    cap = 1 << 20
    while cap < 0.3 * qps * seconds:
        cap <<= 1
    return cap


def predict_speedup(n1_step_ms, per_rank_ms):
    """Strong scaling of ONE pool sharded by chain: the pool's step on one GPU over the slowest rank's step
    (each rank's figure = ITS chains together on one GPU, overlapped as measured).  Not the load share: one GPU
    already runs all chains concurrently, so the heaviest chain's time bounds every N."""
    slow = max([float(t) for t in per_rank_ms] or [0.0])
    return (float(n1_step_ms) / slow) if slow > 0 else None


def expected_digest(key):
    try:
        with open(DIGESTS) as f:
            return json.load(f).get(key)
    except Exception:
        return None


# What `python bench.py --gpus N` re-executes once per rank when no launcher has set WORLD_SIZE (launch_ranks).  The dry
# run of tests/test_bench_line.py points it at its own worker script (gloo + the shim engine) before it calls main().
SELF_CMD = [sys.executable, os.path.abspath(__file__)]


def launch_ranks(n, argv):
    """`python bench.py --gpus N` as ONE command, nobody having set WORLD_SIZE: this process becomes the launcher of N
    ranks of itself — env rendezvous on 127.0.0.1, RANK = LOCAL_RANK = 0 .. N-1, one process per GPU, what
    `python -m torch.distributed.run --nproc-per-node N` would set — relays rank 0's stdout (the ONE JSON line) and
    returns the first non-zero exit code.  So a plain `--gpus 8` can never print an N = 1 number."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), MM_BENCH_LAUNCHED="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # dmabuf IPC: what RCCL needs on this driver
        procs.append(subprocess.Popen(SELF_CMD + list(argv), env=env, stdout=subprocess.PIPE if r == 0 else sys.stderr))
    import threading
    chunks = []
    rd = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read()), daemon=True)
    rd.start()
    # a rank that dies leaves the others in a collective: end them (by their own PIDs) instead of waiting for RCCL's time-out
    while any(p.poll() is None for p in procs):
        if any(p.poll() not in (None, 0) for p in procs):
            for p in procs:
                if p.poll() is None:
                    p.terminate()
            break
        time.sleep(0.05)
    rcs = [p.wait() for p in procs]
    rd.join(timeout=10)
    sys.stdout.write(b"".join(chunks).decode())
    sys.stdout.flush()
    bad = [(r, c) for r, c in enumerate(rcs) if c != 0]
    if bad:
        print("bench.py: rank(s) %s exited non-zero %s" % ([r for r, _ in bad], [c for _, c in bad]), file=sys.stderr)
        return bad[0][1] if bad[0][1] > 0 else 1
    return 0


def main():
    args = parse()
    if "WORLD_SIZE" not in os.environ:
        if args.gpus > 1:
            rc = launch_ranks(args.gpus, sys.argv[1:])
            if rc:
                sys.exit(rc)
            return
    elif int(os.environ["WORLD_SIZE"]) != args.gpus:
        # a launcher that made another number of ranks than the line would claim: refuse, do not print a mislabelled number
        print("bench.py: --gpus %d but WORLD_SIZE=%s — launch one rank per GPU (python -m torch.distributed.run "
              "--nproc-per-node %d ... bench.py --gpus %d), or run `python bench.py --gpus %d` without a launcher and it "
              "starts the ranks itself" % (args.gpus, os.environ["WORLD_SIZE"], args.gpus, args.gpus, args.gpus), file=sys.stderr)
        sys.exit(2)
    import torch
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert torch.cuda.is_available(), "bench.py needs a GPU (there is no CPU fallback)"
    if args.same_device:
        local_rank = 0
    if world > 1 and torch.cuda.device_count() <= local_rank:
        print("bench.py: rank %d wants GPU %d, this node shows %d" % (rank, local_rank, torch.cuda.device_count()), file=sys.stderr)
        sys.exit(3)
    torch.cuda.set_device(local_rank)
    dist = None
    rccl_ranks = 1
    if world > 1 or args.force_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world == 1:                              # --force-dist without a launcher: a rendezvous of one
            import socket
            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                os.environ.setdefault("MASTER_PORT", str(sk.getsockname()[1]))
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        if args.same_device:
            dist.init_process_group("gloo")
            ones = torch.ones(1)
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
            ones = torch.ones(1).cuda()
        dist.all_reduce(ones)                       # the ranks the collective really connected: an all-reduce of ones
        rccl_ranks = int(ones.item())
    coll_dev = "cpu" if args.same_device else "cuda"

    from microservice_matchmaking_amd import Engine, make_config, mode_1v1, mode_team
    from microservice_matchmaking_amd.sharding import (ChainSharding, ShardedSearch, chain_weights, rating_groups,
                                                       tick_digests, union_digest)
    from microservice_matchmaking_amd.synth import ROLE_WEIGHTS_5V5, make_pool

    def workload(mode):
        if mode == "1v1":
            return {"mode": "1v1", "modes": [mode_1v1(window=args.window, region_filter=True)], "pool_kw": {},
                    "bytes_per_pair": BYTES_PER_PAIR, "window": args.window,
                    "describe": lambda n: "1v1, %d players, +-%d rating + region filter (8 regions), %s ratings" % (
                        n, args.window, args.dist)}
        return {"mode": "5v5", "modes": [mode_team(5, 2, 50, (1, 1, 1, 1, 1))], "pool_kw": {"role_weights": ROLE_WEIGHTS_5V5},
                "bytes_per_pair": 12, "window": 50,
                "describe": lambda n: "5v5 team balance, %d players, +-50 rating + 5 roles, %s ratings" % (n, args.dist)}

    wl = workload(args.mode)
    describe = wl["describe"]

    def pow2(n):
        cap = 1
        while cap < max(n, 2):
            cap <<= 1
        return cap

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    pools = {}

    def pool_of(w, n):
        key = (w["mode"], n)
        if key not in pools:
            pools.clear()                                  # one pool at a time (10M players: 80 MB of host arrays)
            pools[key] = make_pool(n, seed=1, dist=args.dist, **w["pool_kw"])
        return pools[key]

    def shared_pool_run(w, n, nranks, r, steps, warmup, timing=True, collective=True):
        """One pool of n players, sharded over nranks by chain; this process plays rank r of them.
        Returns (elapsed of the timed steps, last tick, per-step timers, my digests, the sharding, players
        held, the pool).  collective=False: a single-process run (no barrier with the other ranks)."""
        rating, cons = pool_of(w, n)
        modes = w["modes"]
        cfg1 = make_config(modes, capacity=16, device=local_rank, timing=False)
        sh = ChainSharding(1, cfg1.n_groups, nranks, chain_weights(cfg1, rating, cons)[0])
        grp = rating_groups(cfg1, rating)
        idx = np.nonzero(sh.chain_owner[0][grp.astype(np.int64)] == r)[0]
        timers = {"walk": [], "bucket": [], "copy": [], "step": []}
        digests, last = {}, None
        elapsed = 0.0
        sync = fence if collective else torch.cuda.synchronize
        if idx.size:
            cfg = make_config(modes, capacity=pow2(idx.size), device=local_rank, timing=timing)
            d_rating = torch.from_numpy(rating[idx]).cuda()
            d_cons = torch.from_numpy(cons[idx].view(np.int32)).cuda()
            eng = Engine(cfg)

            def step():
                eng.reset()
                first = eng.enqueue_device(d_rating, d_cons)
                # the match list lands in buffers the binding keeps (views valid until the next tick, as the ABI's list)
                return first, eng.tick(0, reuse=True), dict(eng.last_enqueue_stats)
            for _ in range(warmup):
                step()
        sync()
        t0 = time.perf_counter()
        if idx.size:
            ts = t0
            for _ in range(steps):
                first, last, est = step()
                # (a step ends with its match list on the host: no synchronisation is added to take its time)
                tn = time.perf_counter()
                timers["step"].append((tn - ts) * 1e3)
                ts = tn
                timers["walk"].append(last.stats["walk_ms"])
                timers["bucket"].append(est["bucket_ms"])
                timers["copy"].append(last.stats["copy_ms"])
            torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
        sync()
        if idx.size:
            ids = idx[(last.slots.astype(np.int64) - first) % int(cfg.capacity)] if len(last) else \
                np.zeros(last.slots.shape, np.int64)
            digests = {c: d for c, d in tick_digests(0, cfg.n_groups, ids, last.group).items()
                       if sh.chain_owner[c] == r}
            timers["path"] = eng.path_stats()        # mm_path_stats_get: the launch shapes and fall-backs of the last timed tick
            eng.close()
        return elapsed, last, timers, digests, sh, int(idx.size), (rating, cons)

    def secondary_pool_leg(w, n, steps, traffic_path=None):
        """A whole pool on this GPU beside the main line: value, step, walk, exactness and (1M pools) its roofline."""
        el, l, tm, dg, _, _, _ = shared_pool_run(w, n, 1, 0, steps, 1, collective=False)
        key = workload_key(w["mode"], n, w["window"], args.dist)
        want = expected_digest(key)
        step_ms = el / steps * 1e3
        walk_ms = float(np.mean(tm["walk"]))
        out = {"workload": w["describe"](n), "value": float(l.stats["players_matched"]) * steps / el,
               "unit": "matched players/s", "ms_per_step": step_ms, "steps": steps,
               "passes_max": int(l.stats["passes_max"]), "walk_ms": walk_ms, "pairs_per_step": float(l.stats["pairs"]),
               "exact": (union_digest(dg) == want) if want else None,
               "path": tm.get("path"), "degraded": bool((tm.get("path") or {}).get("degraded")),
               "ms_per_step_min_median_max": [float(np.min(tm["step"])), float(np.median(tm["step"])), float(np.max(tm["step"]))],
               "exactness": {"emission_digest": union_digest(dg), "oracle_digest": want, "key": key}}
        if traffic_path:
            tdet = {}
            traffic, tnote = load_traffic(traffic_path, n, w["mode"], tdet)
            # (the block itself is made at the end of the run, when the kernel boundary has been measured)
            out["_roofline_args"] = (w["mode"], float(l.stats["pairs"]), w["bytes_per_pair"], walk_ms, step_ms, n,
                                     int(l.stats["passes_max"]), traffic, tnote, tdet, tm.get("path"))
        return out

    # The kernel-boundary micro-measurement runs LAST (rank 0): it captures a graph on a side stream of torch's, and that
    # stream stays alive in torch's pool — with it the process holds more streams than the runtime has hardware queues
    # (four), two engines' streams end up sharing one and the concurrent_pools leg measures 87 instead of 152 M/s.
    boundary_us = None
    tdefault = os.path.join(ROOT, "profiles",
                            "traffic_latest.json" if args.mode == "1v1" else "traffic_latest_%s.json" % args.mode)

    # ------------------------------------------------------------------ the main leg
    n = args.players if args.players else (1_000_000 if world == 1 else 10_000_000)
    elapsed, last, timers, digests, sh, n_mine, (rating, cons) = shared_pool_run(wl, n, world, rank, args.steps, args.warmup)
    st = last.stats if last is not None else {"players_matched": 0, "pairs": 0, "passes_max": 0}
    part = {"rank": rank, "elapsed": elapsed, "players": n_mine, "matched_per_step": int(st["players_matched"]),
            "pairs": int(st["pairs"]), "passes_max": int(st["passes_max"]),
            "walk_ms": float(np.mean(timers["walk"])) if timers["walk"] else 0.0,
            "bucket_ms": float(np.mean(timers["bucket"])) if timers["bucket"] else 0.0,
            "copy_ms": float(np.mean(timers["copy"])) if timers["copy"] else 0.0, "digests": digests,
            "step_ms": [float(np.min(timers["step"])), float(np.median(timers["step"])), float(np.max(timers["step"]))]
            if timers["step"] else None,
            "walk_ms_mmm": [float(np.min(timers["walk"])), float(np.median(timers["walk"])), float(np.max(timers["walk"]))]
            if timers["walk"] else None,
            "path": timers.get("path")}
    parts = [part]
    if dist is not None:
        parts = [None] * world
        dist.all_gather_object(parts, part)

    line = None
    if rank == 0:
        slow = max(parts, key=lambda p: p["elapsed"])            # the slowest rank bounds the pool
        elapsed_max = slow["elapsed"]
        matched_step = sum(p["matched_per_step"] for p in parts)
        pairs = float(sum(p["pairs"] for p in parts))
        got = {}
        for p in parts:
            got.update(p["digests"])
        key = workload_key(args.mode, n, wl["window"], args.dist)
        want = expected_digest(key)
        traffic, tnote = (None, "PMC traffic is per single-GPU tick; not collected for the sharded run")
        tdet = {}
        if world == 1:
            traffic, tnote = load_traffic(args.traffic_json or tdefault, n, args.mode, tdet)
        step_ms = elapsed_max / args.steps * 1e3
        line = {
            "metric": "matched players/sec over 1M-player pool" if world == 1 else
                      "matched players/sec over one 10M-player pool sharded across the GPUs",
            "value": matched_step * args.steps / elapsed_max,
            "unit": "matched players/s",
            "n_gpus": world,
            "rccl_ranks": rccl_ranks,
            "collective_backend": "none" if dist is None else ("gloo" if args.same_device else "nccl (RCCL)"),
            "same_device": bool(args.same_device and world > 1),
            "launched_by": "bench.py itself (--gpus N without a launcher)" if os.environ.get("MM_BENCH_LAUNCHED") else
                           ("torch.distributed.run / env" if world > 1 else "single process"),
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": step_ms,
            # the K timed steps one by one (host clock; the slowest rank's): what the mean above hides — runs of this line on
            # one box have differed by 8 %, so a reader can tell a regression from the box
            "ms_per_step_min_median_max": slow["step_ms"],
            "walk_ms_min_median_max": slow["walk_ms_mmm"],
            # which launch shapes the last timed tick of the slowest rank took (mm_path_stats_get) — `degraded`: a fall-back was in
            # force (a kp_rounds stop or its cool-down, kp_rounds off, kt_fc chunk flags that did not come): results identical,
            # the timing not the headline path's
            "path": slow["path"],
            "degraded": bool(any((p.get("path") or {}).get("degraded") for p in parts)),
            "higher_is_better": True,
            "scaling": "weak" if world == 1 else "strong",
            "vs_baseline": None,
            "dtype": "int32",
            "data": "synthetic",
            "config": {
                "workload": describe(n) if world == 1 else
                            "%s — ONE pool sharded across %dxMI355X by (game mode, rating group)" % (describe(n), world),
                "baseline_config": "configs[1]" if (world == 1 and n == 1_000_000 and args.mode == "1v1") else
                                   ("configs[2]" if (world == 1 and n == 1_000_000 and args.mode == "5v5") else
                                    ("configs[3]" if (world > 1 and n == 10_000_000 and args.mode == "1v1") else "custom")),
                "rating_groups": 7,
                "sharding": dict(sh.describe(), **{
                    "collective": "none on the data path (chains never interact: reference lib/application.ex:26-40, "
                                  "lib/models/lobby_state.ex:74-83); RCCL only for barriers and result gathering",
                    "no_halo_allgather": "a chain has ONE open lobby and ONE cursor (lobby_state.ex:90-91); every pass "
                                         "depends on the pass before, so a rating-bucket split of a chain would hop "
                                         "devices once per pass — there is no order-free boundary to exchange "
                                         "(DESIGN.md section 7)",
                    "per_rank": [{"rank": p["rank"], "players": p["players"], "matched_players": p["matched_per_step"],
                                  "ms_per_step": p["elapsed"] / args.steps * 1e3, "passes_max": p["passes_max"]}
                                 for p in parts]}),
                "step": "reset + enqueue(device-resident share) + search to quiescence + match list D2H"},
            "exactness": {"emission_digest": union_digest(got), "oracle_digest": want,
                          "ok": (union_digest(got) == want) if want else None, "key": key,
                          "what": "blake2b over every chain's emission list (global arrival indices, publish order); "
                                  "oracle_digest = the CPU oracle on the same pool (tests/golden/shared_pool_digests.json)"},
            "matched_fraction": matched_step / float(n),
            "pair_evals_per_s": pairs / (slow["walk_ms"] * 1e-3) if slow["walk_ms"] > 0 else None,
            "pairs_per_step": pairs,
            "passes_max": max(p["passes_max"] for p in parts),
            "kernel_ms": {"walk": slow["walk_ms"], "bucket(count+scan+scatter)": slow["bucket_ms"],
                          "d2h+bookkeeping": slow["copy_ms"], "of_rank": slow["rank"]},
            "roofline": None,
        }
        main_roofline_args = (args.mode, pairs, wl["bytes_per_pair"], slow["walk_ms"], step_ms, n,
                              max(p["passes_max"] for p in parts), traffic, tnote, tdet, slow["path"])
        if world == 1 and not args.no_cpu_baseline:
            cfg_cpu = make_config(wl["modes"], capacity=pow2(n), device=local_rank, timing=False)
            line["cpu_baseline"] = cpu_baseline(cfg_cpu, rating, cons, args.mode, budget_s=args.cpu_baseline_seconds)
    del rating, cons

    # ------------------------------------------------------------------ secondary legs (never `value`)
    if not args.no_secondary:
        if world == 1:
            # BASELINE configs[2] (5v5, 1M players, roles + rating) beside the configs[1] headline
            if args.mode == "1v1" and not args.no_cfg3:
                w3 = workload("5v5")
                line["cfg3"] = dict(secondary_pool_leg(w3, n, max(2, args.cfg3_steps),
                                                       os.path.join(ROOT, "profiles", "traffic_latest_5v5.json")),
                                    baseline_config="configs[2]" if n == 1_000_000 else "custom")
            # the same step with the pool handed over in HOST buffers (mm_enqueue: H2D inside the clock) — the
            # PCIe-inclusive rate; never `value`
            if not args.no_pcie:
                rating_h, cons_h = pool_of(wl, n)
                pcfg = make_config(wl["modes"], capacity=pow2(n), device=local_rank, timing=False)
                with Engine(pcfg) as peng:
                    def pstep():
                        peng.reset()
                        peng.enqueue(rating_h, cons_h)
                        return peng.tick(0, reuse=True)
                    pstep()
                    torch.cuda.synchronize()
                    psteps = max(2, min(args.steps, 10))
                    t0 = time.perf_counter()
                    pm = 0
                    for _ in range(psteps):
                        pm += int(pstep().stats["players_matched"])
                    torch.cuda.synchronize()
                    pel = time.perf_counter() - t0
                line["pcie_inclusive"] = {"value": pm / pel, "unit": "matched players/s", "ms_per_step": pel / psteps * 1e3,
                                          "steps": psteps,
                                          "note": "the pool in host memory: mm_enqueue copies rating + cons (8 B a player) to the "
                                                  "device and the handles back inside the step; secondary, never `value`"}
            # cfg-4's pool on ONE GPU: the N=1 point of the strong-scaling curve of `--gpus N`
            n10 = args.shared_players
            if n10 and n10 != n:
                steps10 = max(2, min(args.steps, 8))
                sp = secondary_pool_leg(wl, n10, steps10)
                if not args.no_prediction:
                    # what `--gpus N` will show: every rank's share of this pool run ALONE on this GPU
                    pred = {}
                    for nr in (2, 4, 8):
                        per_rank = []
                        for r in range(nr):
                            el, l, tm, _, shp, held, _ = shared_pool_run(wl, n10, nr, r, 3, 1, collective=False)
                            per_rank.append({"rank": r, "players": held, "ms_per_step": el / 3 * 1e3,
                                             "walk_ms": float(np.mean(tm["walk"])) if tm["walk"] else 0.0,
                                             "passes_max": int(l.stats["passes_max"]) if l is not None else 0,
                                             "chains": [list(c) for c in shp.chains_of(r)]})
                        pred[str(nr)] = {"per_rank": per_rank, "load_share_bound": shp.bound(),
                                         "predicted_speedup": predict_speedup(sp["ms_per_step"],
                                                                              [p["ms_per_step"] for p in per_rank])}
                    sp["sharding_prediction"] = dict(pred, note=(
                        "strong scaling of this pool by (game mode, rating group) chain, measured on ONE GPU: the step of "
                        "the whole pool over the slowest rank's share run alone.  One GPU already runs all chains "
                        "concurrently in the same launches, so the heaviest chain's passes bound every N — "
                        "the load share (load_share_bound) is not the bound (DESIGN.md section 7)"))
                line["shared_pool_n1"] = sp
            if args.concurrent_pools > 1:
                ccfg = make_config(wl["modes"], capacity=pow2(n), device=local_rank, timing=False)

                def pool_inputs(k):
                    r, c = make_pool(n, seed=101 + k, dist=args.dist, **wl["pool_kw"])
                    return torch.from_numpy(r).cuda(), torch.from_numpy(c.view(np.int32)).cuda()

                def pool_digest(m):
                    # a fresh pool's slot handles are its arrival indices (mm_reset + one enqueue)
                    return union_digest(tick_digests(0, ccfg.n_groups, m.slots.astype(np.int64), m.group))

                line["concurrent_pools"] = concurrent_pools(
                    lambda: Engine(ccfg), args.concurrent_pools, max(2, min(args.steps, 20)), pool_inputs, pool_digest,
                    lambda k: expected_digest(workload_key(args.mode, n, wl["window"], args.dist, seed=101 + k)))
        else:
            # the N=1 point of this very pool, on rank 0 alone (the others wait at the barrier): the speed-up of
            # the sharded run is measured inside ONE launch of bench.py, from the gathered per-rank records
            if not args.no_prediction:
                n1 = None
                if rank == 0:
                    el, _, _, _, _, _, _ = shared_pool_run(wl, n, 1, 0, 3, 1, collective=False)
                    n1 = el / 3 * 1e3
                fence()
                if rank == 0:
                    per_rank_ms = [p["elapsed"] / args.steps * 1e3 for p in parts]
                    line["config"]["sharding"]["one_gpu_ms_per_step"] = n1
                    line["config"]["sharding"]["speedup_vs_one_gpu"] = predict_speedup(n1, per_rank_ms)
                    line["config"]["sharding"]["speedup_note"] = (
                        "the whole pool on rank 0's GPU alone (3 steps) over the slowest rank's step of the sharded run; "
                        "bounded by the heaviest chain's passes, not by load_share_bound (DESIGN.md section 7)")
            # weak scaling: every rank its own 1M pool (what round 1 reported as the value)
            r1, c1 = make_pool(args.weak_players, seed=1 + rank, dist=args.dist, **wl["pool_kw"])
            wcfg = make_config(wl["modes"], capacity=pow2(args.weak_players), device=local_rank, timing=False)
            d_r, d_c = torch.from_numpy(r1).cuda(), torch.from_numpy(c1.view(np.int32)).cuda()
            wsteps = max(2, min(args.steps, 20))
            with Engine(wcfg) as weng:
                def wstep():
                    weng.reset()
                    weng.enqueue_device(d_r, d_c)
                    return weng.tick(0)
                wstep()
                fence()
                t0 = time.perf_counter()
                wm = 0
                for _ in range(wsteps):
                    wm += int(wstep().stats["players_matched"])
                fence()
                wel = time.perf_counter() - t0
            tt = torch.tensor([wel, 0.0], device=coll_dev, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            ss = torch.tensor([float(wm)], device=coll_dev, dtype=torch.float64)
            dist.all_reduce(ss, op=dist.ReduceOp.SUM)
            if rank == 0:
                line["weak_scaling"] = {"value": float(ss.item()) / float(tt[0].item()), "unit": "matched players/s",
                                        "pool_per_gpu": args.weak_players, "steps": wsteps,
                                        "ms_per_step": float(tt[0].item()) / wsteps * 1e3,
                                        "note": "every rank its own pool (independent pools of a node: the node-level "
                                                "throughput lever under the reference's partition); secondary, never `value`"}
    pools.clear()

    if not args.no_stream:
        w25 = mode_1v1(window=args.window, region_filter=True)
        mix_w = np.outer([0.7, 0.3], [0.30, 0.10, 0.10, 0.10, 0.10, 0.10, 0.20])   # expected share of every chain
        scap = stream_capacity(args.stream_qps, args.stream_seconds)
        if world == 1 and args.mode == "1v1":
            scfg = make_config([w25], capacity=scap, device=local_rank, timing=False)
            with ShardedSearch(scfg, Engine, 0, 1) as s1:
                line["latency"] = stream_leg(s1, None, 0, 1, args.stream_qps, args.stream_seconds, args.stream_tick_ms,
                                             "1v1 +-%d, region filter" % args.window, "1v1")
            if not args.no_saturation:
                line["latency_saturation"] = saturation_leg(
                    lambda cap: ShardedSearch(make_config([w25], capacity=cap, device=local_rank, timing=False), Engine, 0, 1),
                    args.window, tick_ms=args.stream_tick_ms, seconds=args.saturation_seconds,
                    qps_list=tuple(int(q) for q in args.saturation_qps.split(",")))
        if args.mode == "1v1":
            # BASELINE cfg-5: 70 % 1v1 / 30 % 5v5 (roles as cfg-3), chains = (mode, group) over the ranks
            mcfg = make_config([w25, mode_team(5, 2, 50, (1, 1, 1, 1, 1))], capacity=scap, device=local_rank, timing=False)
            with ShardedSearch(mcfg, Engine, rank, world, mix_w) as sm:
                res = stream_leg(sm, dist, rank, world, args.stream_qps, args.stream_seconds, args.stream_tick_ms,
                                 "70 %% 1v1 +-%d region filter / 30 %% 5v5 +-50 five roles" % args.window, "mixed",
                                 mode_weights=(70, 30), role_weights=ROLE_WEIGHTS_5V5)
                if rank == 0:
                    res["sharding"] = sm.sharding.describe()
                    line["latency_mixed"] = res

    if rank == 0:
        boundary_us = None if args.no_boundary else measure_boundary_us(torch)
        a = main_roofline_args
        probe = None
        if world == 1 and args.mode == "1v1" and not args.no_probe:
            # the predicate tests the pair kernels PHYSICALLY perform in one tick of the headline pool: a second engine whose
            # mm_tuning has pair_tune bit 13 (the counters are atomics — never on the timed engine), one untimed step
            rating_p, cons_p = pool_of(wl, n)
            qcfg = make_config(wl["modes"], capacity=pow2(n), device=local_rank, timing=False)
            with Engine(qcfg, {"pair_tune": 0x2000}) as qeng:
                qeng.enqueue_device(torch.from_numpy(rating_p).cuda(), torch.from_numpy(cons_p.view(np.int32)).cuda())
                qeng.tick(0)
                qs = qeng.path_stats()
            probe = {"tests_all": qs["pair_tested_lo"] + (qs["pair_tested_hi"] << 32),
                     "tests_nx_init": qs["pair_tested_nx_lo"] + (qs["pair_tested_nx_hi"] << 32)}
        line["roofline"] = roofline_block(*a[:8], boundary_us, a[8], a[9], a[10], probe)
        if "cfg3" in line and "_roofline_args" in line["cfg3"]:
            a = line["cfg3"].pop("_roofline_args")
            line["cfg3"]["roofline"] = roofline_block(*a[:8], boundary_us, a[8], a[9], a[10])
        for leg in ("shared_pool_n1",):
            if leg in line:
                line[leg].pop("_roofline_args", None)
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    # The process's stdout (file descriptor 1) belongs to the ONE JSON line.  Libraries print there too — RCCL its banner
    # ("Hostname : ...", "Librccl path : ...") when the `nccl` process group of the N > 1 branch comes up, gloo its connection
    # report — and the driver reads stdout: descriptor 1 goes to stderr for the whole run, the line is written to a copy of
    # the original.  (Found in round 6 by running the RCCL branch for the first time, on one rank: `--force-dist`.)
    sys.stdout.flush()
    _line_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    sys.stdout = _line_out
    main()
    _line_out.flush()
