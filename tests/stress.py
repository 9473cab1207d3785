#!/usr/bin/env python
"""Test infrastructure (it drives the oracle, so it lives under tests/).  Randomised differential stress on a real GPU: HIP engine vs oracle over many seeded
scenarios (pool sizes that hit the LDS-resident walk, the tiled rounds and the hand-over
between them; windows from 0 to wider than the rating span; 1..64 regions; multi-tick with
arrivals and cancels).  Usage: python tests/stress.py [seconds] [seed] [team] [--fuzz-knobs]

--fuzz-knobs (round 6): every scenario's engine is created with a random COMBINATION of the engine's tuning fields
(include/mm_engine.h mm_tuning, passed per engine through mm_engine_create_ex) off their defaults — batch sizes, the
persistent launch shapes on / off / cut short, bounded waits of zero, the test hooks that make a kp_rounds launch stop at a
random iteration and a kt_fc chunk flag never come.  Round 5's tile-length bug needed a stop of kp_rounds to show and no
default-configuration test could see it; one knob at a time found it, combinations are what this draws.  The draw is a
function of the scenario's seed alone (MM_STRESS_ONLY=<seed> replays scenario AND knobs); a failure prints both.

MM_STRESS_ENGINE=emu_small runs the same scenarios without a GPU on the fiber-shim build of the
kernel source with the tiny tile geometry (tests/emu/), pool sizes divided by 16 so that they
land on the same paths (LDS-resident walk / tiled rounds / hand-over; team path / k_walk)."""
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

from helpers import assert_same_state, assert_same_tick  # noqa: E402
from microservice_matchmaking_amd import Engine, cons_make, make_config, mode_1v1, mode_team  # noqa: E402
from oracle.oracle import OracleEngine  # noqa: E402

SCALE = 1
if os.environ.get("MM_STRESS_ENGINE", "gpu") == "emu_small":
    from emu_engine import EmuEngineSmall as Engine  # noqa: E402,F811
    SCALE = 16
elif os.environ.get("MM_STRESS_ENGINE", "gpu") == "emu":      # product geometry under the shim: slow, sizes as on the GPU
    from emu_engine import EmuEngine as Engine  # noqa: E402,F811


def scaled(n):
    return int(n) // SCALE if n >= 100 else int(n)


FUZZ = False          # set by main() from the command line

# field of mm_tuning -> the values a draw picks from (the default is always among them or is what "not drawn" leaves)
PAIR_KNOBS = {
    "pair_persist": [0, 1], "pair_ptiles": [3, 5, 6, 8, 12, 20, 32], "pair_pbatch": [1, 2, 7, 48, 96],
    "pair_batch": [1, 2, 7, 48], "pair_ptimeout_us": [0, 50, 5000], "pair_pinject": [0] + list(range(1, 41)),
    "pair_tiles_max": [10, 20, 40], "pair_tile_fixed": [0, 1], "pair_xcd": [0, 1], "pair_group_min": [0, 4, 64],
    "pair_nxseg": [0, 64, 256, 1024, 2048], "pair_nxstage": [0, 512, 2048],
}
TEAM_KNOBS = {
    "team_batch": [1, 2, 4, 16], "team_f2": [0, 1, 3, 32, 1000], "team_rebuild": [1, 3, 8, 32], "team_emit_max": [1, 2, 8, 32],
    "team_split": [0, 1], "team_fwait": [0, 16, 16384], "team_fix_max": [0, 3, 0xFFFFFFFF], "team_fix_t8": [0, 10, 64],
    "team_fix_t4": [0, 64, 1000], "team_pull_xcd": [0, 1], "team_nowait": [0, 1, 2, 3, 7], "team_late": [0, 1, 6, 80],
    "team_late0": [0, 512, 10 ** 7], "team_cap": [8, 64, 512, 4096],
}
COMMON_KNOBS = {"results_early": [0, 1], "results_tail": [0, 1], "look_poll": [0, 1]}


def draw_tuning(seed, tables):
    """None without --fuzz-knobs; else {field: value}: every field of `tables` drawn with probability 0.35."""
    if not FUZZ:
        return None
    krng = np.random.default_rng([int(seed), 0x6B6E6F62])       # its own stream: the scenario is the same with and without
    t = {}
    for table in tables + [COMMON_KNOBS]:
        for name, menu in table.items():
            if krng.random() < 0.35:
                t[name] = int(menu[krng.integers(0, len(menu))])
    if krng.integers(0, 50) == 0:
        t["force_generic"] = 1
    return t


def random_team_mode(rng):
    """A team mode the config validator accepts: teams * team_size <= 16, quotas sum to team_size."""
    teams = int(rng.choice([2, 2, 2, 3, 4]))
    team_size = int(rng.integers(1 if teams > 2 else 2, 16 // teams + 1))
    team_size = min(team_size, 8)
    n_roles = int(rng.integers(1, min(team_size, 5) + 1))
    quota = np.ones(n_roles, np.int64)
    for _ in range(team_size - n_roles):
        quota[rng.integers(0, n_roles)] += 1
    window = int(rng.choice([5, 25, 50, 150, 600, 10 ** 6]))
    return mode_team(team_size, teams, window, tuple(int(q) for q in quota),
                     region_filter=bool(rng.integers(0, 3) == 0), party_filter=bool(rng.integers(0, 5) == 0))


def team_main(budget, seed0):
    """Team modes only: long chains take mm_team.inc, short ones and cancel ticks k_walk."""
    t_end = time.time() + budget
    n_done = 0
    k = 0
    only = os.environ.get("MM_STRESS_ONLY")
    while time.time() < t_end:
        seed = seed0 * 100003 + k
        k += 1
        if only:
            if k > 1:
                break
            seed = int(only)
        rng = np.random.default_rng(seed)
        modes = [random_team_mode(rng)]
        nr = modes[0]["n_roles"]
        cfg = make_config(modes, capacity=1 << 18, timing=False)
        regions = int(rng.choice([1, 2, 8])) if modes[0]["region_filter"] else 1
        parties = 3 if modes[0]["party_filter"] else 1
        lo = int(rng.choice([0, 0, 1000, 2400]))
        hi = int(rng.choice([1499, 2600, 5000, 5000]))
        if hi <= lo:
            hi = lo + 600
        w = rng.random(nr) + 0.05
        w /= w.sum()
        sizes = [scaled(rng.choice([3000, 20000, 60000, 120000]))] + \
                [scaled(rng.choice([0, 100, 5000, 30000])) for _ in range(int(rng.integers(0, 4)))]
        tuning = draw_tuning(seed, [TEAM_KNOBS])
        tag = "team seed %d mode=%s ratings=[%d,%d] sizes=%s tuning=%s" % (seed, modes[0], lo, hi, sizes, tuning)
        if os.environ.get("MM_STRESS_VERBOSE"):
            print(tag, flush=True)
        with (Engine(cfg, tuning) if tuning else Engine(cfg)) as a, OracleEngine(cfg) as b:
            live = np.zeros(0, np.uint32)
            for j, n in enumerate(sizes):
                rating = rng.integers(lo, hi + 1, size=n).astype(np.int32)
                cons = cons_make(0, rng.integers(0, regions, size=n), rng.integers(0, parties, size=n),
                                 rng.choice(nr, size=n, p=w))
                sa, sb = a.enqueue(rating, cons), b.enqueue(rating, cons)
                assert np.array_equal(sa, sb), tag
                live = np.concatenate([live, sa])
                if live.size > 10 and rng.integers(0, 4) == 0:
                    cs = rng.choice(live, size=max(1, live.size // 50), replace=False)
                    a.cancel(cs)
                    b.cancel(cs)
                    live = np.setdiff1d(live, cs)
                ma, mb = a.tick(0), b.tick(0)
                assert_same_tick(ma, mb, tag + " tick %d" % j)
                live = np.setdiff1d(live, ma.slots.ravel())
                assert_same_state(a, b, cfg, tag)
        n_done += 1
    print("gpu_stress team%s: %d scenarios ok (seeds %d..%d)" % (" --fuzz-knobs" if FUZZ else "", n_done, seed0 * 100003, seed0 * 100003 + k - 1))


def main(argv=None):
    global FUZZ
    argv = list(sys.argv[1:] if argv is None else argv)
    FUZZ = "--fuzz-knobs" in argv
    ARGV = [a for a in argv if not a.startswith("--")]
    budget = float(ARGV[0]) if len(ARGV) > 0 else 30.0
    seed0 = int(ARGV[1]) if len(ARGV) > 1 else 1
    if len(ARGV) > 2 and ARGV[2] == "team":
        return team_main(budget, seed0)
    t_end = time.time() + budget
    n_done = 0
    k = 0
    only = os.environ.get("MM_STRESS_ONLY")              # one scenario by its seed (what a failure's message names)
    while time.time() < t_end:
        seed = seed0 * 100003 + k
        k += 1
        if only:
            if k > 1:
                break
            seed = int(only)
        rng = np.random.default_rng(seed)
        window = int(rng.choice([0, 1, 3, 10, 25, 60, 200, 1000, 10 ** 6]))
        regions = int(rng.choice([1, 2, 4, 8, 64]))
        party = bool(rng.integers(0, 4) == 0)
        modes = [mode_1v1(window=window, region_filter=regions > 1, party_filter=party)]
        if rng.integers(0, 3) == 0:
            modes.append(mode_team(2, 2, 300, (1, 1)))
        cfg = make_config(modes, capacity=1 << 19, timing=False)
        lo = int(rng.choice([0, 0, 1000, 2400]))
        hi = int(rng.choice([1499, 2600, 5000, 5000]))
        if hi <= lo:
            hi = lo + 600
        sizes = [scaled(rng.choice([50, 3000, 20000, 70000, 150000, 260000]))] + \
                [scaled(rng.choice([0, 100, 5000, 40000])) for _ in range(int(rng.integers(0, 4)))]
        tuning = draw_tuning(seed, [PAIR_KNOBS] + ([TEAM_KNOBS] if len(modes) > 1 else []))
        tag = "seed %d w=%d regions=%d party=%d modes=%d ratings=[%d,%d] sizes=%s tuning=%s" % (
            seed, window, regions, party, len(modes), lo, hi, sizes, tuning)
        if os.environ.get("MM_STRESS_VERBOSE"):
            print(tag, flush=True)                      # (a crash inside the library leaves no assertion message behind)
        with (Engine(cfg, tuning) if tuning else Engine(cfg)) as a, OracleEngine(cfg) as b:
            live = np.zeros(0, np.uint32)
            for j, n in enumerate(sizes):
                rating = rng.integers(lo, hi + 1, size=n).astype(np.int32)
                mode = rng.integers(0, len(modes), size=n) if len(modes) > 1 else np.zeros(n, np.int64)
                role = np.where(mode == 1, rng.integers(0, 2, size=n), 0)
                cons = cons_make(mode, rng.integers(0, regions, size=n), rng.integers(0, 3 if party else 1, size=n), role)
                sa, sb = a.enqueue(rating, cons), b.enqueue(rating, cons)
                assert np.array_equal(sa, sb), tag
                live = np.concatenate([live, sa])
                if live.size > 10 and rng.integers(0, 3) == 0:
                    cs = rng.choice(live, size=max(1, live.size // 50), replace=False)
                    a.cancel(cs)
                    b.cancel(cs)
                    live = np.setdiff1d(live, cs)
                for md in range(len(modes)):
                    ma, mb = a.tick(md), b.tick(md)
                    assert_same_tick(ma, mb, tag + " tick %d mode %d" % (j, md))
                    live = np.setdiff1d(live, ma.slots.ravel())
                assert_same_state(a, b, cfg, tag)
        n_done += 1
    print("gpu_stress%s: %d scenarios ok (seeds %d..%d)" % (" --fuzz-knobs" if FUZZ else "", n_done, seed0 * 100003, seed0 * 100003 + k - 1))


if __name__ == "__main__":
    main()
