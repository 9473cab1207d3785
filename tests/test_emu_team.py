"""The team path (mm_team.inc: kt_build / kt_f / kt_chase / kt_emit) under the CPU shim, built
with a small geometry (tests/emu/libmm_engine_emu_small.so: TT_MIN=64, TT_SCAN_CAP=24) so that
chains of a few hundred players take it and kt_f's scan cap is hit.  Every tick is compared with
the oracle: lobbies, their order, team layout, scores, pass numbers, counters, queue depths and
the stored lobbies.  Logic tests only; the parity gate is test_gpu_parity.py."""
import numpy as np
import pytest

from emu_engine import EmuEngine, EmuEngineSmall
from helpers import assert_same_state, assert_same_tick, random_scenario
from microservice_matchmaking_amd._abi import cons_make
from microservice_matchmaking_amd.config import make_config, mode_1v1, mode_team

W5 = [0.15, 0.15, 0.30, 0.30, 0.10]      # SURVEY.md section 8(d) cfg-3 role weights


def ticks(oracle_cls, engine_cls, mode, n, seed, n_ticks=3, regions=1, lo=0, hi=5000, weights=None,
          capacity=16384):
    cfg = make_config([mode], capacity=capacity)
    rng = np.random.default_rng(seed)
    nr = cfg.modes[0].n_roles
    total = 0
    with engine_cls(cfg) as a, oracle_cls(cfg) as b:
        for k in range(n_ticks):
            nn = n if k == 0 else n // 3
            rating = rng.integers(lo, hi + 1, size=nn).astype(np.int32)
            cons = cons_make(0, rng.integers(0, regions, size=nn), 0, rng.choice(nr, size=nn, p=weights))
            assert np.array_equal(a.enqueue(rating, cons), b.enqueue(rating, cons))
            ma, mb = a.tick(0), b.tick(0)
            assert_same_tick(ma, mb, "tick %d" % k)
            assert_same_state(a, b, cfg, "tick %d" % k)
            total += len(ma)
    return total


def test_team_5v5_roles_and_stored_lobbies(oracle_cls):
    """cfg-3's mode: the scarce role starves the lobbies, every chain ends a tick with a stored
    lobby that the next tick fills from the head of the queue."""
    assert ticks(oracle_cls, EmuEngineSmall, mode_team(5, 2, 50, (1, 1, 1, 1, 1)), 2500, seed=1, weights=W5) > 50


def test_team_2v2_single_role(oracle_cls):
    assert ticks(oracle_cls, EmuEngineSmall, mode_team(2, 2, 100, (2,)), 2000, seed=2) > 500


def test_team_three_teams_region_filter(oracle_cls):
    assert ticks(oracle_cls, EmuEngineSmall, mode_team(2, 3, 400, (1, 1), region_filter=True), 2400, seed=3,
                 regions=3) > 300


def test_team_uneven_quota_dense_window(oracle_cls):
    """Quota (2,1,1), a window that takes almost everybody: lobbies fill within a few players,
    passes end with the lobby filled by the last queued player."""
    assert ticks(oracle_cls, EmuEngineSmall, mode_team(4, 2, 2000, (2, 1, 1)), 2400, seed=4,
                 weights=[0.2, 0.4, 0.4]) > 150


def test_team_8v8_full_lobby_width(oracle_cls):
    """16 seats = MM_MAX_LOBBY; a pass that ends exactly on a fill (no open lobby left)."""
    assert ticks(oracle_cls, EmuEngineSmall, mode_team(8, 2, 800, (2, 2, 2, 2)), 2400, seed=6) > 150


def test_team_narrow_window_scan_cap(oracle_cls):
    """+-20 in one rating group: most lobbies need more than TT_SCAN_CAP entries of a sub-queue,
    kt_chase resolves them with the wave-wide scan (filled and not filled)."""
    assert ticks(oracle_cls, EmuEngineSmall, mode_team(5, 2, 20, (1, 1, 1, 1, 1)), 1500, seed=5, n_ticks=2,
                 lo=0, hi=1400, weights=W5) > 10


@pytest.mark.parametrize("every", ["1", "3", "100000"])
def test_team_rebuild_cadence(oracle_cls, monkeypatch, every):
    """MM_TEAM_REBUILD: role sub-queues rebuilt in every pass, every third, or only in the first two passes of
    a tick — in between, players that leave are tombstones in them and the ranks count the entries of the
    last rebuild.  Same lobbies (stored lobbies, several ticks)."""
    monkeypatch.setenv("MM_TEAM_REBUILD", every)
    assert ticks(oracle_cls, EmuEngineSmall, mode_team(5, 2, 60, (1, 1, 1, 1, 1)), 1500, seed=8, n_ticks=3,
                 lo=0, hi=1400, weights=W5) > 10


def test_team_cancel_ticks(oracle_cls):
    """Ticks with pending cancels on the team path: k_purge, then kt_init judges the head against
    the stale lobby and filters it (the head sits out the first pass or is seated), and a stored
    lobby whose lower teams a cancel emptied moves its anchor while it fills.  A 1v1 mode rides
    along on the same engine (pair path)."""
    cfg = make_config([mode_team(3, 2, 300, (1, 1, 1)), mode_1v1(window=80, region_filter=True)], capacity=16384)
    rng = np.random.default_rng(21)
    with EmuEngineSmall(cfg) as a, oracle_cls(cfg) as b:
        random_scenario(rng, cfg, a, b, n_rounds=5, batch=2500, cancel_frac=0.02)
    rng = np.random.default_rng(22)
    with EmuEngineSmall(cfg) as a, oracle_cls(cfg) as b:
        random_scenario(rng, cfg, a, b, n_rounds=4, batch=2500, cancel_frac=0.0)


@pytest.mark.parametrize("seed,mode,cancel_frac", [
    (3, mode_team(1, 4, 30, (1,), region_filter=True), 0.2),       # four teams of one: the anchor moves often
    (4, mode_team(5, 3, 5000, (1, 1, 2, 1)), 0.05),
    (8, mode_team(2, 3, 5000, (2,), region_filter=True), 0.2),
    (11, mode_team(2, 2, 150, (1, 1)), 0.05),
])
def test_team_cancel_ticks_lobby_shapes(oracle_cls, seed, mode, cancel_frac):
    """Heavy cancelling on team modes with three and four teams: stale lobbies, heads that the
    stale lobby takes or rejects, anchors that move down to an emptied team."""
    cfg = make_config([mode], capacity=16384)
    rng = np.random.default_rng(seed)
    with EmuEngineSmall(cfg) as a, oracle_cls(cfg) as b:
        random_scenario(rng, cfg, a, b, n_rounds=5, batch=1500, cancel_frac=cancel_frac)


def test_team_anchor_moves_to_the_emptied_first_team(oracle_cls):
    """Hand-built: the stored lobby is team 1: [A, C], team 2: [B]; A and C cancel.  The next
    tick's head meets the stale lobby (anchor A: rejected, it sits out the pass), the filter
    leaves team 2: [B] with B the anchor.  D fits B and is seated in the empty team 1, which makes
    D the anchor: E (fits D, not B) is taken, F (fits B, not D) is not; G completes the lobby a
    tick later."""
    cfg = make_config([mode_team(2, 2, 100, (2,))], capacity=4096)
    others = 1000 + 5 * np.arange(70)                    # one rating group, nobody near 500
    with EmuEngineSmall(cfg) as a, oracle_cls(cfg) as b:
        def both(fn):
            ra, rb = fn(a), fn(b)
            return ra, rb
        r1 = np.concatenate([[500, 510, 520], others]).astype(np.int32)
        sa, sb = both(lambda e: e.enqueue(r1, cons_make(np.zeros(r1.size))))
        assert np.array_equal(sa, sb)
        ma, mb = both(lambda e: e.tick(0))
        assert_same_tick(ma, mb, "tick 1")
        assert_same_state(a, b, cfg, "tick 1")
        slots, teams = a.lobby_state(0, 0)
        assert slots.tolist() == [int(sa[0]), int(sa[2]), int(sa[1])] and teams.tolist() == [0, 0, 1]
        both(lambda e: e.cancel(np.asarray([sa[0], sa[2]], np.uint32)))
        r2 = np.asarray([600, 690, 450], np.int32)       # D, E, F
        s2, _ = both(lambda e: e.enqueue(r2, cons_make(np.zeros(3))))
        ma, mb = both(lambda e: e.tick(0))
        assert_same_tick(ma, mb, "tick 2")
        assert_same_state(a, b, cfg, "tick 2")
        slots, teams = a.lobby_state(0, 0)
        assert slots.tolist() == [int(s2[0]), int(sa[1]), int(s2[1])] and teams.tolist() == [0, 1, 1]
        s3, _ = both(lambda e: e.enqueue(np.asarray([650], np.int32), cons_make(np.zeros(1))))
        ma, mb = both(lambda e: e.tick(0))
        assert_same_tick(ma, mb, "tick 3")
        assert len(ma) >= 1 and ma.slots[0].tolist() == [int(s2[0]), int(s3[0]), int(sa[1]), int(s2[1])]
        assert_same_state(a, b, cfg, "tick 3")


def test_team_extreme_ratings_fall_back(oracle_cls):
    """A chain whose rating span does not fit the packed key is walked by the generic kernel,
    its neighbours by the team path, in the same tick."""
    cfg = make_config([mode_team(2, 2, 60, (1, 1))], capacity=8192)
    rng = np.random.default_rng(5)
    n = 1200
    rating = rng.integers(3000, 3400, size=n).astype(np.int32)
    rating[::200] = np.asarray([2**31 - 1, -2**31, 2**31 - 5, -2**31 + 7, 2**30, -2**30], np.int32)
    grp = np.full(n, 4, np.uint8)             # host routed them all to one group ...
    rating2 = rng.integers(1500, 2000, size=600).astype(np.int32)      # ... and these go where they belong
    with EmuEngineSmall(cfg) as a, oracle_cls(cfg) as b:
        cons = cons_make(0, 0, 0, rng.integers(0, 2, size=n))
        assert np.array_equal(a.enqueue(rating, cons, grp), b.enqueue(rating, cons, grp))
        cons2 = cons_make(0, 0, 0, rng.integers(0, 2, size=600))
        assert np.array_equal(a.enqueue(rating2, cons2), b.enqueue(rating2, cons2))
        assert_same_tick(a.tick(0), b.tick(0), "extreme")
        assert_same_state(a, b, cfg)


def test_team_short_chains_stay_with_k_walk(oracle_cls):
    """Product geometry (TT_MIN=4096): one long chain takes the team path, the short ones of the
    same tick are walked by k_walk."""
    cfg = make_config([mode_team(2, 2, 150, (1, 1))], capacity=16384)
    rng = np.random.default_rng(8)
    n = 5200
    rating = np.where(rng.random(n) < 0.8, rng.integers(0, 1500, size=n), rng.integers(1500, 5001, size=n)).astype(np.int32)
    cons = cons_make(0, 0, 0, rng.integers(0, 2, size=n))
    with EmuEngine(cfg) as a, oracle_cls(cfg) as b:
        assert np.array_equal(a.enqueue(rating, cons), b.enqueue(rating, cons))
        assert_same_tick(a.tick(0), b.tick(0), "mixed")
        assert_same_state(a, b, cfg)


def test_team_path_on_a_starving_stream(oracle_cls):
    """The regime cfg-5's 60 s run ends in (helpers.run_starving_team_stream; the GPU twin is in test_gpu_parity.py):
    long chains, a handful of lobbies per tick, cancels every other tick — on the small geometry (TT_MIN = 64)."""
    from helpers import run_starving_team_stream
    per, depth = run_starving_team_stream(EmuEngineSmall, oracle_cls, preload=9000, ticks=14, per_tick=60, cancels=6,
                                          capacity=1 << 14)
    assert depth.min() >= 64 and sum(per) > 0


@pytest.mark.parametrize("late,late0", [("0", "512"), ("1000", "512"), ("1", "10000000"), ("6", "10000000")])
def test_kt_late_takes_the_chains_over_at_any_point(oracle_cls, monkeypatch, late, late0):
    """kt_late (one persistent workgroup per chain for the passes that seat a handful of lobbies) must give the pass
    kernels' results wherever the hand-over happens: MM_TEAM_LATE=1000 hands every chain over after the first batch
    (two passes), 0 never does; MM_TEAM_LATE0 = arrivals since the mode's last tick up to which kt_late walks a tick from
    its FIRST pass — with 10^7 every tick starts there, and the big first ticks make it hand the chains back after a pass
    that seated more than 4 x MM_TEAM_LATE + 32 lobbies (the pass kernels go on behind a rebuild of the sub-queues).
    Cancel ticks, stored lobbies and the starving stream included."""
    from helpers import run_starving_team_stream
    monkeypatch.setenv("MM_TEAM_LATE", late)
    monkeypatch.setenv("MM_TEAM_LATE0", late0)
    assert ticks(oracle_cls, EmuEngineSmall, mode_team(5, 2, 50, (1, 1, 1, 1, 1)), 2500, seed=3, weights=W5) > 50
    assert ticks(oracle_cls, EmuEngineSmall, mode_team(2, 3, 500, (2,)), 1500, seed=4, regions=2) > 50
    per, depth = run_starving_team_stream(EmuEngineSmall, oracle_cls, preload=6000, ticks=8, per_tick=80, cancels=9,
                                          capacity=1 << 14)
    assert sum(per) > 0
    cfg = make_config([mode_team(3, 2, 400, (2, 1), region_filter=True)], capacity=1 << 13)
    rng = np.random.default_rng(5)
    with EmuEngineSmall(cfg) as a, oracle_cls(cfg) as b:
        random_scenario(rng, cfg, a, b, n_rounds=4, batch=1500, cancel_frac=0.05)


def test_kt_late_hands_a_chain_back_after_a_rich_pass(oracle_cls, monkeypatch):
    """kt_late from the first pass (MM_TEAM_LATE0) on a pool whose first pass seats more lobbies than `late_bail`
    (4 x MM_TEAM_LATE + 32): it stops at the pass boundary, the sub-queues are rebuilt (kt_late leaves no tombstones)
    and the pass kernels finish the tick — same lobbies, order, counters and state as the oracle's."""
    monkeypatch.setenv("MM_TEAM_LATE", "1")
    monkeypatch.setenv("MM_TEAM_LATE0", "10000000")
    assert ticks(oracle_cls, EmuEngineSmall, mode_team(5, 2, 50, (1, 1, 1, 1, 1)), 12000, seed=9, n_ticks=2,
                 weights=[0.2] * 5, capacity=1 << 15) > 300


@pytest.mark.parametrize("f2,split", [("0", "1"), ("2", "1"), ("1000", "1"), ("1000", "0")])
def test_every_launch_shape_of_a_pass(oracle_cls, monkeypatch, f2, split):
    """A pass is kt_f | kt_f2 | kt_chase (the first MM_TEAM_F2 passes of a tick; the stored lobby's fill in kt_f's launch or
    in kt_chase's: MM_TEAM_SPLIT) or kt_f, the chase and the emitters in ONE launch (kt_fc, the passes from MM_TEAM_F2 on) —
    every combination must give the oracle's ticks: cancel ticks, stored lobbies, the scan cap, the starving stream.  (Rounds
    2-5 also had kt_f | kt_chase | kt_emit one behind the other, MM_TEAM_LIVE=0 / MM_TEAM_FUSED=0: taken out in round 6.
    Under the shim the workgroups of a launch run one after another, the ones that wait last: this tests the roles' logic
    and the publish / take protocol, not their overlap — test_gpu_parity.py does that.)"""
    from helpers import run_starving_team_stream
    monkeypatch.setenv("MM_TEAM_F2", f2)
    monkeypatch.setenv("MM_TEAM_SPLIT", split)         # the stored lobby's fill from the head of the queue in kt_f's launch (the passes with kt_f2) or in kt_chase's
    monkeypatch.setenv("MM_TEAM_LATE", "0")
    assert ticks(oracle_cls, EmuEngineSmall, mode_team(5, 2, 50, (1, 1, 1, 1, 1)), 2500, seed=13, weights=W5) > 50
    assert ticks(oracle_cls, EmuEngineSmall, mode_team(2, 3, 500, (2,)), 1500, seed=14, regions=2) > 50
    if f2 == "0":                                        # kt_fc from the first pass: the scan cap and the starving stream too
        assert ticks(oracle_cls, EmuEngineSmall, mode_team(4, 2, 30, (2, 1, 1)), 3000, seed=15, lo=0, hi=900) > 20   # narrow window: the scan cap
        per, depth = run_starving_team_stream(EmuEngineSmall, oracle_cls, preload=6000, ticks=6, per_tick=80, cancels=9,
                                              capacity=1 << 14)
        assert sum(per) > 0
    cfg = make_config([mode_team(3, 2, 400, (2, 1), region_filter=True)], capacity=1 << 13)
    rng = np.random.default_rng(16)
    with EmuEngineSmall(cfg) as a, oracle_cls(cfg) as b:
        random_scenario(rng, cfg, a, b, n_rounds=4, batch=1500, cancel_frac=0.05)


@pytest.mark.parametrize("nowait", ["1", "2", "3"])
def test_a_chaser_that_gets_no_flag_looks_the_lobby_up_itself(oracle_cls, monkeypatch, nowait):
    """kt_fc: the chaser waits a bounded number of polls (MM_TEAM_FWAIT) for the flag of a kt_f chunk — a workgroup that
    may not have found a CU yet when somebody else's kernels hold them — and then looks the lobby up itself, which is what it
    does for anchors beyond kt_f's horizon anyway; the emitter is told (TV_LOOKED in vis[]) not to trust kt_f's record of
    that anchor, which may be in the making.  MM_TEAM_NOWAIT=n is the test hook: the flag of every n-th chunk never comes
    (1: no flag at all — the chaser walks the whole tick by itself).  A late chunk is never a failed tick."""
    from helpers import run_starving_team_stream
    monkeypatch.setenv("MM_TEAM_NOWAIT", nowait)
    monkeypatch.setenv("MM_TEAM_F2", "0")               # kt_fc from the first pass
    monkeypatch.setenv("MM_TEAM_LATE", "0")
    assert ticks(oracle_cls, EmuEngineSmall, mode_team(5, 2, 50, (1, 1, 1, 1, 1)), 2500, seed=17, weights=W5) > 50
    assert ticks(oracle_cls, EmuEngineSmall, mode_team(2, 3, 500, (2,)), 1500, seed=18, regions=2) > 50
    assert ticks(oracle_cls, EmuEngineSmall, mode_team(4, 2, 30, (2, 1, 1)), 3000, seed=19, lo=0, hi=900) > 20
    per, depth = run_starving_team_stream(EmuEngineSmall, oracle_cls, preload=6000, ticks=6, per_tick=80, cancels=9,
                                          capacity=1 << 14)
    assert sum(per) > 0


@pytest.mark.parametrize("fixmax,t8,t4", [("0", "10", "64"), ("2", "10", "64"), ("100000", "10", "64"), ("100000", "0", "0"), ("100000", "0", "1000")])
def test_lobbies_mended_or_looked_up_from_scratch(oracle_cls, monkeypatch, fixmax, t8, t4):
    """kt_f keeps the lobby an anchor opens on record; when members of it have left, it replaces them (the first entries
    that fit behind the role's last member in the role's sub-queue: kt_build leaves sqi for players that left as well) —
    unless the chunk has more than MM_TEAM_FIXMAX such replacements to make, in which case its anchors are looked up from
    scratch, as anchors without a record always are.  0: never mend (round 3's kt_f); 2: both ways in one tick; 100000:
    always mend.  The small geometry has records of 5-bit distances (TF_FAR_BITS): replacements that do not fit the record."""
    from helpers import run_starving_team_stream
    monkeypatch.setenv("MM_TEAM_FIXMAX", fixmax)
    monkeypatch.setenv("MM_TEAM_FIXT8", t8)            # replacements a wave has above which each gets eight / four lanes instead of sixteen
    monkeypatch.setenv("MM_TEAM_FIXT4", t4)
    monkeypatch.setenv("MM_TEAM_LATE", "0")
    assert ticks(oracle_cls, EmuEngineSmall, mode_team(5, 2, 50, (1, 1, 1, 1, 1)), 2500, seed=31, weights=W5) > 50
    assert ticks(oracle_cls, EmuEngineSmall, mode_team(4, 2, 30, (2, 1, 1)), 3000, seed=33, n_ticks=2, lo=0, hi=900) > 20   # narrow window: the scan cap
    if fixmax != "100000" or t8 != "10":
        return
    # the defaults: more shapes (the product geometry: test_team_short_chains_stay_with_k_walk, and the gpu tier)
    assert ticks(oracle_cls, EmuEngineSmall, mode_team(2, 3, 500, (2,)), 1500, seed=32, regions=2) > 50
    per, depth = run_starving_team_stream(EmuEngineSmall, oracle_cls, preload=6000, ticks=6, per_tick=80, cancels=9,
                                          capacity=1 << 14)
    assert sum(per) > 0
    cfg = make_config([mode_team(3, 2, 400, (2, 1), region_filter=True)], capacity=1 << 13)
    rng = np.random.default_rng(35)
    with EmuEngineSmall(cfg) as a, oracle_cls(cfg) as b:
        random_scenario(rng, cfg, a, b, n_rounds=4, batch=1500, cancel_frac=0.05)


def test_path_stats_count_the_chunk_flags_that_did_not_come(oracle_cls, monkeypatch):
    """mm_path_stats_get on the team path: the launches of a tick by kind, and — with the test hook that keeps every third
    chunk's flag from coming (MM_TEAM_NOWAIT=3) — the anchors the chaser looked up itself, with `degraded` set: a late
    chunk is a slower pass, and the tick's record says so ("What's weak" 8: 'same for kt_fc chunk time-outs')."""
    monkeypatch.setenv("MM_TEAM_F2", "0")               # kt_fc from the first pass
    monkeypatch.setenv("MM_TEAM_LATE", "0")
    cfg = make_config([mode_team(5, 2, 50, (1, 1, 1, 1, 1))], capacity=1 << 13)
    from microservice_matchmaking_amd.synth import ROLE_WEIGHTS_5V5, make_pool
    rating, cons = make_pool(2500, seed=17, role_weights=ROLE_WEIGHTS_5V5)
    rating = (rating % 1400).astype(np.int32)           # everybody in the first rating group: one chain of ten chunks
    out = {}
    for nowait in ("0", "3"):
        monkeypatch.setenv("MM_TEAM_NOWAIT", nowait)
        with EmuEngineSmall(cfg) as a, oracle_cls(cfg) as b:
            assert np.array_equal(a.enqueue(rating, cons), b.enqueue(rating, cons))
            ma, mb = a.tick(0), b.tick(0)
            assert_same_tick(ma, mb, "path stats, nowait " + nowait)
            out[nowait] = a.path_stats()
    clean, late = out["0"], out["3"]
    for ps in (clean, late):
        assert ps["paths"] == 4 and ps["mode"] == 0 and ps["host_looks"] >= 2 and ps["crit_passes"] == 0
        assert ps["team_fc_launches"] > 0 and ps["team_f_launches"] == 0 and ps["team_build_launches"] >= 2
        assert ps["pair_rounds_launches"] == ps["pair_round_launches"] == 0
    assert clean["team_flags_late"] == 0 and clean["degraded"] == 0
    assert late["team_flags_late"] > 0 and late["team_flags_late_total"] == late["team_flags_late"] and late["degraded"] == 1
