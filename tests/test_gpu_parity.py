"""Parity tests proper: the HIP engine through the C ABI vs the oracle, on a real MI355X.

Bit-exact on assignments (slots per lobby, team order, emission order, rating group, pass),
counters (pairs, scanned, passes, pool sizes) and lobby/queue state; scores within 1e-6
(north_star tolerance for the floating rating-delta)."""
import numpy as np
import pytest

from helpers import assert_same_state, assert_same_tick, load_golden, random_scenario, run_golden_case
from microservice_matchmaking_amd import cons_make, make_config, mode_1v1, mode_team
from microservice_matchmaking_amd.synth import ROLE_WEIGHTS_5V5, make_pool

pytestmark = pytest.mark.gpu
GOLD = load_golden()
SCORE_TOL = 1e-6


@pytest.fixture(scope="module")
def gpu_cls():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU; there is no CPU fallback"
    from microservice_matchmaking_amd import Engine
    return Engine


@pytest.mark.parametrize("case", GOLD["cases"], ids=[c["name"] for c in GOLD["cases"]])
def test_gpu_golden_case(gpu_cls, case):
    run_golden_case(gpu_cls, case)


MODE_SETS = {
    "1v1": [mode_1v1(window=60)],
    "1v1_region": [mode_1v1(window=40, region_filter=True)],
    "5v5_roles": [mode_team(5, 2, 400, (1, 1, 1, 1, 1))],
    "mixed": [mode_1v1(window=40, region_filter=True), mode_team(2, 2, 300, (1, 1)),
              mode_team(3, 2, 400, (3,), party_filter=True)],
    "3teams": [mode_team(2, 3, 500, (2,))],
    "4x4": [mode_team(4, 4, 800, (2, 1, 1))],
}


@pytest.mark.parametrize("mset", sorted(MODE_SETS))
@pytest.mark.parametrize("seed", [1, 2])
def test_gpu_random_scenarios(gpu_cls, oracle_cls, mset, seed):
    cfg = make_config(MODE_SETS[mset], capacity=1 << 16)
    rng = np.random.default_rng(100 * seed + len(mset))
    with gpu_cls(cfg) as a, oracle_cls(cfg) as b:
        random_scenario(rng, cfg, a, b, n_rounds=5, batch=6000, cancel_frac=0.04)


def check_properties(cfg, mode, rating, cons, m, n_enq):
    """Size-independent invariants of a tick (hold for any Mode R run)."""
    mc = cfg.modes[mode]
    L = mc.teams * mc.team_size
    flat = m.slots.ravel().astype(np.int64)
    assert flat.size == len(m) * L
    assert np.unique(flat).size == flat.size, "a player appears in two lobbies"
    assert flat.min(initial=0) >= 0 and flat.max(initial=0) < n_enq
    r = rating[flat].reshape(-1, L).astype(np.int64)
    c = cons[flat].reshape(-1, L)
    # everyone within the window of the anchor (team 1's first player)
    assert (np.abs(r - r[:, :1]) <= mc.window).all()
    if mc.flags & 1:
        assert (((c >> 4) & 0xFF) == ((c[:, :1] >> 4) & 0xFF)).all()
    # all of one rating group, groups in emission order, passes ascending inside a group
    assert (np.diff(m.group.astype(np.int64)) >= 0).all()
    same = np.diff(m.group.astype(np.int64)) == 0
    assert (np.diff(m.pass_.astype(np.int64))[same] >= 0).all()
    # score = spread of team sums / team_size
    sums = r.reshape(-1, mc.teams, mc.team_size).sum(axis=2)
    want = ((sums.max(axis=1) - sums.min(axis=1)).astype(np.float32) / np.float32(mc.team_size))
    assert np.allclose(m.score, want, atol=SCORE_TOL, rtol=0)
    # role quotas per team
    roles = ((c >> 16) & 0xF).reshape(-1, mc.teams, mc.team_size)
    for rr in range(mc.n_roles):
        assert ((roles == rr).sum(axis=2) == mc.role_quota[rr]).all()
    assert m.stats["pool_before"] == m.stats["pool_after"] + len(m) * L


@pytest.mark.parametrize("n,dist", [(1000, "uniform"), (65536, "uniform"), (65536, "normal"),
                                    (1000000, "uniform"), (1000000, "normal")])
def test_gpu_1v1_region_pool(gpu_cls, oracle_cls, n, dist):
    """BASELINE cfg-1/cfg-2: 1v1, +-25 rating, region filter, seeded synthetic pool."""
    cfg = make_config([mode_1v1(window=25 if n > 1000 else 50, region_filter=n > 1000)], capacity=1 << 20)
    rating, cons = make_pool(n, seed=1, dist=dist)
    with gpu_cls(cfg) as a, oracle_cls(cfg) as b:
        assert np.array_equal(a.enqueue(rating, cons), b.enqueue(rating, cons))
        ma, mb = a.tick(0), b.tick(0)
        assert_same_tick(ma, mb, "1v1 n=%d %s" % (n, dist), SCORE_TOL)
        assert_same_state(a, b, cfg)
        check_properties(cfg, 0, rating, cons, ma, n)
        # idempotence: a quiescent pool stays quiescent
        ma2, mb2 = a.tick(0), b.tick(0)
        assert len(ma2) == 0 and len(mb2) == 0
        assert_same_tick(ma2, mb2, "second tick")


@pytest.mark.parametrize("n,dist", [(65536, "uniform"), (1000000, "uniform"), (1000000, "normal")])
def test_gpu_5v5_pool(gpu_cls, oracle_cls, n, dist):
    """BASELINE cfg-3: 5v5 team balance, role + rating constraints."""
    cfg = make_config([mode_team(5, 2, 50, (1, 1, 1, 1, 1))], capacity=1 << 20)
    rating, cons = make_pool(n, seed=2, dist=dist, role_weights=ROLE_WEIGHTS_5V5)
    with gpu_cls(cfg) as a, oracle_cls(cfg) as b:
        a.enqueue(rating, cons)
        b.enqueue(rating, cons)
        ma, mb = a.tick(0), b.tick(0)
        assert_same_tick(ma, mb, "5v5 n=%d %s" % (n, dist), SCORE_TOL)
        assert_same_state(a, b, cfg)
        check_properties(cfg, 0, rating, cons, ma, n)


def test_gpu_streaming_ticks_mixed_modes(gpu_cls, oracle_cls):
    """cfg-5 in miniature: batches arrive between ticks, 70/30 1v1/5v5, cancels trickle in."""
    cfg = make_config([mode_1v1(window=25, region_filter=True), mode_team(5, 2, 50, (1, 1, 1, 1, 1))],
                      capacity=1 << 18)
    rng = np.random.default_rng(42)
    with gpu_cls(cfg) as a, oracle_cls(cfg) as b:
        live = []
        for tick in range(12):
            n = 20000
            rating, cons = make_pool(n, seed=100 + tick, mode_weights=(70, 30), role_weights=ROLE_WEIGHTS_5V5)
            # 1v1 players carry role 0
            is1 = (cons & 0xF) == 0
            cons = np.where(is1, cons & ~np.uint32(0xF << 16), cons).astype(np.uint32)
            sa, sb = a.enqueue(rating, cons), b.enqueue(rating, cons)
            assert np.array_equal(sa, sb)
            live.extend(sa.tolist())
            if tick % 3 == 2:
                cs = rng.choice(np.asarray(live, dtype=np.uint32), size=500, replace=False)
                a.cancel(cs)
                b.cancel(cs)
                gone = set(cs.tolist())
                live = [s for s in live if s not in gone]
            for mode in (0, 1):
                ma, mb = a.tick(mode), b.tick(mode)
                assert_same_tick(ma, mb, "stream tick %d mode %d" % (tick, mode), SCORE_TOL)
                gone = set(ma.slots.ravel().tolist())
                live = [s for s in live if s not in gone]
            assert_same_state(a, b, cfg, "tick %d" % tick)


def test_gpu_team_path_on_a_starving_stream(gpu_cls, oracle_cls):
    """What the 60 s run of cfg-5 turns into (profiles/r03_bench_stream60*.json: 5v5 backlog 905k players): every 5v5
    chain on the team path (>= 4096 players), a handful of lobbies per tick, cancels trickling in."""
    from helpers import run_starving_team_stream
    per, depth = run_starving_team_stream(gpu_cls, oracle_cls)
    assert depth.min() >= 4096, depth                          # all seven chains take the team path
    assert 0 < np.median(per) <= 40 and min(per[5:]) <= 5 * 7, per      # a handful of lobbies per chain and tick


def test_gpu_team_path_on_a_starving_stream_pass_kernels_only(gpu_cls, oracle_cls, monkeypatch):
    """The same stream without kt_late and without kt_f2 (MM_TEAM_LATE=0, MM_TEAM_F2=0): every pass of every tick is ONE
    kt_fc launch — chasers, kt_f's chunks and emitters side by side — and every tick ends on a pass that seats nobody,
    which the chaser finishes in microseconds while kt_f's workgroups are still being dispatched (the end of a chain
    beside a running kt_f: where a workgroup must stay or leave as a whole).  100 ticks, cancels every other one."""
    from helpers import run_starving_team_stream
    monkeypatch.setenv("MM_TEAM_LATE", "0")
    monkeypatch.setenv("MM_TEAM_F2", "0")
    per, depth = run_starving_team_stream(gpu_cls, oracle_cls, preload=120_000, ticks=100, per_tick=300, cancels=15, seed=12)
    assert depth.min() >= 4096 and sum(per) > 0, (depth, per)


def test_gpu_device_resident_enqueue(gpu_cls, oracle_cls):
    """mm_enqueue_device: inputs already in HBM (the benchmark path) == host-pointer path."""
    import torch
    cfg = make_config([mode_1v1(window=25, region_filter=True)], capacity=1 << 18)
    rating, cons = make_pool(200000, seed=9)
    d_r = torch.from_numpy(rating).cuda()
    d_c = torch.from_numpy(cons.view(np.int32)).cuda()
    with gpu_cls(cfg) as a, oracle_cls(cfg) as b:
        first = a.enqueue_device(d_r, d_c)
        assert first == 0 and a.last_enqueue_stats["accepted"] == 200000
        b.enqueue(rating, cons)
        assert_same_tick(a.tick(0), b.tick(0), "device enqueue", SCORE_TOL)
        a.reset()
        b.reset()
        # after a reset the engine is reusable and deterministic
        a.enqueue_device(d_r, d_c)
        b.enqueue(rating, cons)
        assert_same_tick(a.tick(0), b.tick(0), "after reset", SCORE_TOL)


def test_gpu_edge_cases(gpu_cls, oracle_cls):
    cfg = make_config([mode_1v1(window=50)], capacity=4096)
    with gpu_cls(cfg) as a, oracle_cls(cfg) as b:
        assert len(a.tick(0)) == 0                                   # empty pool
        a.enqueue(np.zeros(0, np.int32), np.zeros(0, np.uint32))     # empty batch
        for e in (a, b):
            e.enqueue(np.asarray([1000], np.int32), cons_make([0]))  # single player
        assert_same_tick(a.tick(0), b.tick(0), "single")
        for e in (a, b):                                             # all-same-rating ties
            e.enqueue(np.full(999, 1000, np.int32), cons_make(np.zeros(999)))
        assert_same_tick(a.tick(0), b.tick(0), "ties")
        for e in (a, b):                                             # zero feasible pairs
            e.enqueue((np.arange(40) * 101 % 1400).astype(np.int32) + 3000 * 0, cons_make(np.zeros(40)))
        assert_same_tick(a.tick(0), b.tick(0), "sparse")
        assert_same_state(a, b, cfg)
        # extreme ratings: |delta| overflows int32 unless computed carefully
        for e in (a, b):
            e.enqueue(np.asarray([2**31 - 1, -2**31, 2**31 - 1, -2**31 + 10], np.int32), cons_make(np.zeros(4)))
        assert_same_tick(a.tick(0), b.tick(0), "extreme ratings")
        assert_same_state(a, b, cfg)


@pytest.mark.parametrize("window,regions,party", [(0, 1, False), (5, 8, False), (400, 64, False),
                                                  (100000, 1, False), (30, 4, True)])
def test_gpu_pair_path_variants(gpu_cls, oracle_cls, window, regions, party):
    """The 1v1 pair path (tiled rounds + LDS-resident walk) across predicate shapes: exact-rating
    windows, a window wider than the rating span, many / no regions, the party filter."""
    cfg = make_config([mode_1v1(window=window, region_filter=regions > 1, party_filter=party)], capacity=1 << 19)
    rating, cons = make_pool(300000, seed=21, n_regions=regions, party_max=3 if party else 1)
    with gpu_cls(cfg) as a, oracle_cls(cfg) as b:
        assert np.array_equal(a.enqueue(rating, cons), b.enqueue(rating, cons))
        ma, mb = a.tick(0), b.tick(0)
        assert_same_tick(ma, mb, "w=%d r=%d" % (window, regions), SCORE_TOL)
        assert_same_state(a, b, cfg)
        check_properties(cfg, 0, rating, cons, ma, 300000)


def test_gpu_pair_path_multi_tick_with_cancels(gpu_cls, oracle_cls):
    """Survivors, the carried anchor and new arrivals over several ticks; a cancel tick in the
    middle is walked by the generic kernel and hands its state back to the pair path."""
    cfg = make_config([mode_1v1(window=25, region_filter=True)], capacity=1 << 20)
    rng = np.random.default_rng(3)
    with gpu_cls(cfg) as a, oracle_cls(cfg) as b:
        live = np.zeros(0, np.uint32)
        for k, n in enumerate([400000, 150000, 5000, 250000]):
            rating, cons = make_pool(n, seed=40 + k, dist="normal" if k % 2 else "uniform")
            sa, sb = a.enqueue(rating, cons), b.enqueue(rating, cons)
            assert np.array_equal(sa, sb)
            live = np.concatenate([live, sa])
            if k == 2:
                cs = rng.choice(live, size=3000, replace=False)
                a.cancel(cs)
                b.cancel(cs)
                live = np.setdiff1d(live, cs)
            ma, mb = a.tick(0), b.tick(0)
            assert_same_tick(ma, mb, "tick %d" % k, SCORE_TOL)
            assert_same_state(a, b, cfg, "tick %d" % k)
            live = np.setdiff1d(live, ma.slots.ravel())


TEAM_VARIANTS = {
    "2v2_one_role": (mode_team(2, 2, 100, (2,)), None, 1),
    "3teams_region": (mode_team(2, 3, 400, (1, 1), region_filter=True), (50, 50), 4),
    "4v4_uneven_quota": (mode_team(4, 2, 150, (2, 1, 1)), (20, 40, 40), 1),
    "8v8_full_width": (mode_team(8, 2, 300, (2, 2, 2, 2)), (25, 25, 25, 25), 1),
    "5v5_narrow_party": (mode_team(5, 2, 20, (1, 1, 1, 1, 1), party_filter=True), (15, 15, 30, 30, 10), 1),
}


@pytest.mark.parametrize("name", sorted(TEAM_VARIANTS))
def test_gpu_team_path_variants(gpu_cls, oracle_cls, name):
    """The team path (mm_team.inc) across lobby shapes: three ticks with arrivals (stored lobbies
    carry over), then a cancel tick (generic kernel) and a tick after it."""
    mode, role_w, regions = TEAM_VARIANTS[name]
    cfg = make_config([mode], capacity=1 << 19)
    rng = np.random.default_rng(5)
    with gpu_cls(cfg) as a, oracle_cls(cfg) as b:
        live = np.zeros(0, np.uint32)
        for k, n in enumerate([200000, 60000, 3000, 40000, 80000]):
            rating, cons = make_pool(n, seed=60 + k, dist="normal" if k == 1 else "uniform", n_regions=regions,
                                     role_weights=role_w, party_max=3 if mode["party_filter"] else 1)
            sa, sb = a.enqueue(rating, cons), b.enqueue(rating, cons)
            assert np.array_equal(sa, sb)
            live = np.concatenate([live, sa])
            if k == 3:
                cs = rng.choice(live, size=2000, replace=False)
                a.cancel(cs)
                b.cancel(cs)
                live = np.setdiff1d(live, cs)
            ma, mb = a.tick(0), b.tick(0)
            assert_same_tick(ma, mb, "%s tick %d" % (name, k), SCORE_TOL)
            assert_same_state(a, b, cfg, "%s tick %d" % (name, k))
            live = np.setdiff1d(live, ma.slots.ravel())


@pytest.mark.parametrize("cap", [1, 16, 100000])
def test_gpu_team_scan_horizon(gpu_cls, oracle_cls, monkeypatch, cap):
    """MM_TEAM_CAP: with a horizon of one sub-queue entry every lobby is worked out by kt_chase's
    wave-wide scan; with none every scan runs to the end of its sub-queue.  Same lobbies."""
    monkeypatch.setenv("MM_TEAM_CAP", str(cap))
    cfg = make_config([mode_team(5, 2, 50, (1, 1, 1, 1, 1))], capacity=1 << 17)
    rating, cons = make_pool(100000, seed=9, role_weights=ROLE_WEIGHTS_5V5)
    with gpu_cls(cfg) as a, oracle_cls(cfg) as b:
        assert np.array_equal(a.enqueue(rating, cons), b.enqueue(rating, cons))
        assert_same_tick(a.tick(0), b.tick(0), "cap %d" % cap, SCORE_TOL)
        assert_same_state(a, b, cfg)


@pytest.mark.parametrize("every", [1, 3, 100000])
def test_gpu_team_rebuild_cadence(gpu_cls, oracle_cls, monkeypatch, every):
    """MM_TEAM_REBUILD: the role sub-queues rebuilt in every pass (no tombstones at all), every third, or only
    in the first two passes of a tick (everybody who leaves afterwards is a tombstone until the tick ends).
    Two ticks with arrivals in between.  Same lobbies."""
    monkeypatch.setenv("MM_TEAM_REBUILD", str(every))
    cfg = make_config([mode_team(5, 2, 50, (1, 1, 1, 1, 1))], capacity=1 << 18)
    with gpu_cls(cfg) as a, oracle_cls(cfg) as b:
        for k, n in enumerate([150000, 40000]):
            rating, cons = make_pool(n, seed=21 + k, role_weights=ROLE_WEIGHTS_5V5)
            assert np.array_equal(a.enqueue(rating, cons), b.enqueue(rating, cons))
            assert_same_tick(a.tick(0), b.tick(0), "rebuild every %d, tick %d" % (every, k), SCORE_TOL)
            assert_same_state(a, b, cfg)


def test_gpu_team_members_beyond_the_record(gpu_cls, oracle_cls):
    """kt_f keeps every anchor's lobby on record as 16-bit distances from the anchor.  Here the players of
    the second role all queue 70 000 positions behind the first anchors, so no lobby fits the record for a
    lobby fits the record in any of the 500 passes (those anchors are looked up again every pass): a pass
    seats one such lobby, the next anchor's stays open and is filled from the head of the queue in the pass
    after.  Same lobbies, same order."""
    n0, n1 = 70000, 2000
    cfg = make_config([mode_team(2, 2, 5000, (1, 1))], groups=[(0, 5000, "all")], capacity=1 << 17)
    rating = np.full(n0 + n1, 2500, np.int32)
    cons = cons_make(0, 0, 0, np.concatenate([np.zeros(n0, np.uint32), np.ones(n1, np.uint32)]))
    with gpu_cls(cfg) as a, oracle_cls(cfg) as b:
        assert np.array_equal(a.enqueue(rating, cons), b.enqueue(rating, cons))
        ma, mb = a.tick(0), b.tick(0)
        assert len(ma) == n1 // 2 and ma.stats["passes_max"] >= n1 // 4
        assert_same_tick(ma, mb, "far members", SCORE_TOL)
        assert_same_state(a, b, cfg)


def test_gpu_restart_from_a_snapshot(gpu_cls, oracle_cls):
    """mm_snapshot / mm_restore on the device: the engine is destroyed and rebuilt from its
    snapshot three times (cancels pending across two of them); pools large enough for the pair
    and team paths."""
    from test_snapshot import restart_script
    restart_script(gpu_cls, oracle_cls, n=120000, seed=5, capacity=1 << 19)


def test_gpu_1v1_10m_pool(gpu_cls, oracle_cls):
    """BASELINE cfg-4 on one device: 10M players, 1v1 +-25 + region filter (chains of 1-3M players,
    hundreds of tiles per chain).  On a node the same chains spread over the ranks by rating group
    (sharding.GroupSharding), each rank running exactly this code on its groups."""
    n = 10_000_000
    cfg = make_config([mode_1v1(window=25, region_filter=True)], capacity=1 << 24)
    rating, cons = make_pool(n, seed=4)
    with gpu_cls(cfg) as a, oracle_cls(cfg) as b:
        sa = a.enqueue(rating, cons)
        assert sa[0] == 0 and sa[-1] == n - 1
        b.enqueue(rating, cons)
        ma, mb = a.tick(0), b.tick(0)
        assert_same_tick(ma, mb, "10M", SCORE_TOL)
        assert_same_state(a, b, cfg)
        check_properties(cfg, 0, rating, cons, ma, n)


def test_gpu_5v5_10m_pool(gpu_cls, oracle_cls):
    """Ten times BASELINE cfg-3 on one device: chains of 1-3M players (thousands of chunks per
    chain for the team path's kernels)."""
    n = 10_000_000
    cfg = make_config([mode_team(5, 2, 50, (1, 1, 1, 1, 1))], capacity=1 << 24)
    rating, cons = make_pool(n, seed=6, role_weights=ROLE_WEIGHTS_5V5)
    with gpu_cls(cfg) as a, oracle_cls(cfg) as b:
        a.enqueue(rating, cons)
        b.enqueue(rating, cons)
        ma, mb = a.tick(0), b.tick(0)
        assert_same_tick(ma, mb, "5v5 10M", SCORE_TOL)
        assert_same_state(a, b, cfg)
        check_properties(cfg, 0, rating, cons, ma, n)


def test_gpu_randomised_stress_short(gpu_cls, monkeypatch):
    """Ten seconds of tests/stress.py: seeded random pools / predicates / multi-tick scripts
    with arrivals and cancels, every tick bit-exact against the oracle."""
    import importlib.util
    import os
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "stress.py")
    spec = importlib.util.spec_from_file_location("gpu_stress", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    mod.main(["10", "3"])


@pytest.mark.parametrize("what", ["pair", "team"])
def test_gpu_randomised_stress_with_fuzzed_knobs(gpu_cls, what, capsys):
    """Twenty seconds each of tests/stress.py --fuzz-knobs (round 6): every scenario's engine is created with a random
    COMBINATION of mm_tuning fields off their defaults — batch sizes, kp_rounds on / off / stopped at a random iteration by
    the test hook, bounded waits of zero, kt_fc chunk flags that never come — and every tick is bit-exact against the oracle.
    Round 5's tile-length bug was invisible to every default-configuration test for three rounds;
    this is the tier the driver runs.  The soak (tens of thousands of scenarios): profiles/r06_stress_fuzz_knobs.txt."""
    import importlib.util
    import os
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "stress.py")
    spec = importlib.util.spec_from_file_location("gpu_stress_fuzz", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    mod.main(["20", "6" if what == "pair" else "7"] + (["team"] if what == "team" else []) + ["--fuzz-knobs"])
    out = capsys.readouterr().out
    assert "--fuzz-knobs" in out and "scenarios ok" in out, out


def test_gpu_pair_second_route_level_forced(gpu_cls, oracle_cls, monkeypatch):
    """kp_group (the second level of the route, by default only for chains of 64+ tiles: the 10M pool) on every tiled
    chain of a 300k pool — MM_PAIR_GROUP=4: groups at 8192-position tiles while the chains shrink through all the
    batches that still use the longest tile — over three ticks with arrivals and cancels."""
    monkeypatch.setenv("MM_PAIR_GROUP", "4")
    cfg = make_config([mode_1v1(window=25, region_filter=True)], capacity=1 << 20)
    rng = np.random.default_rng(12)
    with gpu_cls(cfg) as a, oracle_cls(cfg) as b:
        live = np.zeros(0, np.uint32)
        for tick in range(3):
            rating, cons = make_pool(600000 if tick == 0 else 150000, seed=30 + tick)
            sa, sb = a.enqueue(rating, cons), b.enqueue(rating, cons)
            assert np.array_equal(sa, sb)
            live = np.concatenate([live, sa])
            if tick:
                cs = rng.choice(live, size=3000, replace=False)
                a.cancel(cs)
                b.cancel(cs)
            ma, mb = a.tick(0), b.tick(0)
            assert_same_tick(ma, mb, "group route tick %d" % tick, SCORE_TOL)
            assert_same_state(a, b, cfg, "group route tick %d" % tick)
            live = np.setdiff1d(live, ma.slots.ravel())


@pytest.mark.parametrize("f2,split", [("0", "1"), ("1000", "1"), ("1000", "0"), ("3", "1")])
def test_gpu_team_every_launch_shape_of_a_pass(gpu_cls, oracle_cls, monkeypatch, f2, split):
    """The ways a pass of the team path is launched — kt_f | kt_f2 | kt_chase (the first MM_TEAM_F2 passes; the stored
    lobby's fill in kt_f's launch or in kt_chase's: MM_TEAM_SPLIT) and kt_f, the chase and the emitters in one launch
    (kt_fc, the chasers taking F chunk by chunk while kt_f is still at work) — on the device, where the workgroups of a
    launch really overlap: 200k-player 5v5 pool, then arrivals + cancels, and a dense 3 x 2 mode.  (kt_f | kt_chase |
    kt_emit one behind the other, rounds 2-5's MM_TEAM_LIVE=0 / MM_TEAM_FUSED=0, was taken out in round 6.)"""
    monkeypatch.setenv("MM_TEAM_F2", f2)
    monkeypatch.setenv("MM_TEAM_SPLIT", split)         # the stored lobby's fill in kt_f's launch, beside its chunks (the passes with kt_f2), or in kt_chase's
    cfg = make_config([mode_team(5, 2, 50, (1, 1, 1, 1, 1))], capacity=1 << 19)
    rng = np.random.default_rng(17)
    with gpu_cls(cfg) as a, oracle_cls(cfg) as b:
        live_slots = np.zeros(0, np.uint32)
        for tick in range(4):
            rating, cons = make_pool(200000 if tick == 0 else 3000, seed=50 + tick, role_weights=ROLE_WEIGHTS_5V5)
            sa, sb = a.enqueue(rating, cons), b.enqueue(rating, cons)
            assert np.array_equal(sa, sb)
            live_slots = np.concatenate([live_slots, sa])
            if tick >= 2:
                cs = rng.choice(live_slots, size=200, replace=False)
                a.cancel(cs)
                b.cancel(cs)
            ma, mb = a.tick(0), b.tick(0)
            assert_same_tick(ma, mb, "launch shape tick %d" % tick, SCORE_TOL)
            assert_same_state(a, b, cfg, "launch shape tick %d" % tick)
            live_slots = np.setdiff1d(live_slots, ma.slots.ravel())
    cfg = make_config([mode_team(2, 3, 400, (1, 1), region_filter=True)], capacity=1 << 18)
    with gpu_cls(cfg) as a, oracle_cls(cfg) as b:
        for tick in range(2):
            rating, cons = make_pool(120000 if tick == 0 else 20000, seed=60 + tick, n_regions=3, role_weights=(3, 2))
            assert np.array_equal(a.enqueue(rating, cons), b.enqueue(rating, cons))
            assert_same_tick(a.tick(0), b.tick(0), "launch shape 3x2 tick %d" % tick, SCORE_TOL)
            assert_same_state(a, b, cfg, "launch shape 3x2 tick %d" % tick)


@pytest.mark.parametrize("env", [{"MM_TEAM_NOWAIT": "3"}, {"MM_TEAM_FWAIT": "0"}, {"MM_TEAM_FWAIT": "0", "MM_TEAM_EMIT_MAX": "1"}])
def test_gpu_team_chaser_without_the_flag_of_a_chunk(gpu_cls, oracle_cls, monkeypatch, env):
    """kt_fc's chaser waits a bounded number of polls for the flag of a kt_f chunk (MM_TEAM_FWAIT; a workgroup that has
    found no CU yet because somebody else's kernels hold them) and then looks the lobby up itself and tells the emitter not
    to trust kt_f's record of that anchor (TV_LOOKED) — a late chunk is a slower pass, never a failed tick.  On the device,
    where the chunk's workgroup really writes its records while the emitter collects the lobby again: MM_TEAM_FWAIT=0 gives
    up on every flag that is not up at the first look, MM_TEAM_NOWAIT=3 never sees the flag of every third chunk."""
    monkeypatch.setenv("MM_TEAM_F2", "0")
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    cfg = make_config([mode_team(5, 2, 50, (1, 1, 1, 1, 1))], capacity=1 << 18)
    rng = np.random.default_rng(23)
    with gpu_cls(cfg) as a, oracle_cls(cfg) as b:
        live_slots = np.zeros(0, np.uint32)
        for tick in range(3):
            rating, cons = make_pool(100000 if tick == 0 else 3000, seed=70 + tick, role_weights=ROLE_WEIGHTS_5V5)
            sa, sb = a.enqueue(rating, cons), b.enqueue(rating, cons)
            assert np.array_equal(sa, sb)
            live_slots = np.concatenate([live_slots, sa])
            if tick >= 1:
                cs = rng.choice(live_slots, size=150, replace=False)
                a.cancel(cs)
                b.cancel(cs)
            ma, mb = a.tick(0), b.tick(0)
            assert_same_tick(ma, mb, "flagless tick %d" % tick, SCORE_TOL)
            assert_same_state(a, b, cfg, "flagless tick %d" % tick)
            live_slots = np.setdiff1d(live_slots, ma.slots.ravel())
            # mm_path_stats_get: the flags that did not come are counted and the tick says it ran on a fall-back
            ps = a.path_stats()
            assert ps["paths"] & 4 and ps["team_fc_launches"] > 0 and ps["team_flags_late"] <= ps["team_flags_late_total"]
            if "MM_TEAM_NOWAIT" in env and tick == 0:
                assert ps["team_flags_late"] > 0 and ps["degraded"] == 1, ps
            assert ps["degraded"] == (1 if ps["team_flags_late"] else 0), ps
    cfg = make_config([mode_team(2, 3, 400, (1, 1), region_filter=True)], capacity=1 << 17)
    with gpu_cls(cfg) as a, oracle_cls(cfg) as b:
        rating, cons = make_pool(60000, seed=73, n_regions=3, role_weights=(3, 2))
        assert np.array_equal(a.enqueue(rating, cons), b.enqueue(rating, cons))
        assert_same_tick(a.tick(0), b.tick(0), "flagless 3x2", SCORE_TOL)
        assert_same_state(a, b, cfg, "flagless 3x2")


@pytest.mark.parametrize("fixmax", ["0", "24", "100000"])
def test_gpu_team_lobbies_mended_or_looked_up_from_scratch(gpu_cls, oracle_cls, monkeypatch, fixmax):
    """kt_f replaces the members a recorded lobby has lost (MM_TEAM_FIXMAX: how many such replacements a chunk makes
    before it looks its anchors up from scratch instead) — 0 is round 3's kt_f, 100000 always mends, 24 does both in one
    tick.  200k-player 5v5 pool, then arrivals + cancels; a dense 3 x 2 mode (most anchors lose a member every pass)."""
    monkeypatch.setenv("MM_TEAM_FIXMAX", fixmax)
    cfg = make_config([mode_team(5, 2, 50, (1, 1, 1, 1, 1))], capacity=1 << 19)
    rng = np.random.default_rng(29)
    with gpu_cls(cfg) as a, oracle_cls(cfg) as b:
        live_slots = np.zeros(0, np.uint32)
        for tick in range(3):
            rating, cons = make_pool(200000 if tick == 0 else 3000, seed=80 + tick, role_weights=ROLE_WEIGHTS_5V5)
            sa, sb = a.enqueue(rating, cons), b.enqueue(rating, cons)
            assert np.array_equal(sa, sb)
            live_slots = np.concatenate([live_slots, sa])
            if tick >= 1:
                cs = rng.choice(live_slots, size=200, replace=False)
                a.cancel(cs)
                b.cancel(cs)
            ma, mb = a.tick(0), b.tick(0)
            assert_same_tick(ma, mb, "mend tick %d" % tick, SCORE_TOL)
            assert_same_state(a, b, cfg, "mend tick %d" % tick)
            live_slots = np.setdiff1d(live_slots, ma.slots.ravel())
    cfg = make_config([mode_team(2, 3, 400, (1, 1), region_filter=True)], capacity=1 << 18)
    with gpu_cls(cfg) as a, oracle_cls(cfg) as b:
        for tick in range(2):
            rating, cons = make_pool(120000 if tick == 0 else 20000, seed=83 + tick, n_regions=3, role_weights=(3, 2))
            assert np.array_equal(a.enqueue(rating, cons), b.enqueue(rating, cons))
            assert_same_tick(a.tick(0), b.tick(0), "mend 3x2 tick %d" % tick, SCORE_TOL)
            assert_same_state(a, b, cfg, "mend 3x2 tick %d" % tick)


@pytest.mark.parametrize("late,late0", [("1000", "512"), ("2", "100000000")])
def test_gpu_team_late_kernel_forced(gpu_cls, oracle_cls, monkeypatch, late, late0):
    """kt_late on the device far beyond its default reach: MM_TEAM_LATE=1000 hands every chain over after the first
    two passes of every tick; MM_TEAM_LATE0=10^8 starts every tick in it (and the rich first passes of the big tick
    make it hand the chains back: late_bail = 4 x 2 + 32 lobbies).  200k-player 5v5 pool, then arrivals + cancels."""
    monkeypatch.setenv("MM_TEAM_LATE", late)
    monkeypatch.setenv("MM_TEAM_LATE0", late0)
    cfg = make_config([mode_team(5, 2, 50, (1, 1, 1, 1, 1))], capacity=1 << 19)
    rng = np.random.default_rng(13)
    with gpu_cls(cfg) as a, oracle_cls(cfg) as b:
        live = np.zeros(0, np.uint32)
        for tick in range(4):
            rating, cons = make_pool(200000 if tick == 0 else 3000, seed=40 + tick, role_weights=ROLE_WEIGHTS_5V5)
            sa, sb = a.enqueue(rating, cons), b.enqueue(rating, cons)
            assert np.array_equal(sa, sb)
            live = np.concatenate([live, sa])
            if tick >= 2:
                cs = rng.choice(live, size=200, replace=False)
                a.cancel(cs)
                b.cancel(cs)
            ma, mb = a.tick(0), b.tick(0)
            assert_same_tick(ma, mb, "kt_late forced tick %d" % tick, SCORE_TOL)
            assert_same_state(a, b, cfg, "kt_late forced tick %d" % tick)
            live = np.setdiff1d(live, ma.slots.ravel())



@pytest.mark.gpu
@pytest.mark.parametrize("env", [{}, {"MM_PAIR_PERSIST": "0"}, {"MM_PAIR_PINJECT": "3"}, {"MM_PAIR_PINJECT": "1"},
                                 {"MM_PAIR_PTIMEOUT_US": "0"}, {"MM_PAIR_PBATCH": "7"}, {"MM_PAIR_PTILES": "5"}],
                         ids=lambda e: ",".join("%s=%s" % kv for kv in sorted(e.items())) or "default")
def test_gpu_pair_rounds_stop_and_go_on(gpu_cls, oracle_cls, monkeypatch, env):
    """kp_rounds (several passes per launch, a flag barrier among a chain's workgroups) and the ways it ends: the batch
    runs out, the longest chain yields for its compaction, a workgroup declares a stop (MM_PAIR_PINJECT: tile 1 of every
    chain, behind the given iteration — 1: before anything has been walked), a wait runs out at once
    (MM_PAIR_PTIMEOUT_US=0: whoever is first at a barrier gives up).  Every stop commits the chain and the host walks
    on with one launch per pass (kp_round): same lobbies, same order as the oracle's, tick after tick with cancels."""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    cfg = make_config([mode_1v1(window=25, region_filter=True)], capacity=1 << 19)
    rng = np.random.default_rng(23)
    with gpu_cls(cfg) as a, oracle_cls(cfg) as b:
        live_slots = np.zeros(0, np.uint32)
        for tick in range(3):
            rating, cons = make_pool(300000 if tick == 0 else 20000, seed=70 + tick)
            sa, sb = a.enqueue(rating, cons), b.enqueue(rating, cons)
            assert np.array_equal(sa, sb)
            live_slots = np.concatenate([live_slots, sa])
            if tick >= 1:
                cs = rng.choice(live_slots, size=300, replace=False)
                a.cancel(cs)
                b.cancel(cs)
            ma, mb = a.tick(0), b.tick(0)
            assert_same_tick(ma, mb, "rounds %s tick %d" % (env, tick), SCORE_TOL)
            assert_same_state(a, b, cfg, "rounds %s tick %d" % (env, tick))
            live_slots = np.setdiff1d(live_slots, ma.slots.ravel())
            # mm_path_stats_get: what used to be one MM_PAIR_DEBUG line on stderr is in the tick's record
            ps = a.path_stats()
            stops = ps["pair_stops_timeout"] + ps["pair_stops_xcd"] + ps["pair_stops_inject"]
            assert ps["paths"] & 2 and ps["crit_passes"] == ma.stats["passes_max"]
            assert ps["crit_rounds_passes"] + ps["crit_round_passes"] + ps["crit_late_passes"] == ps["crit_passes"]
            if tick == 0:
                assert stops == ps["pair_stops_total"]             # every stop counted once (not again at the tick's later looks)
                if not env or "MM_PAIR_PBATCH" in env or "MM_PAIR_PTILES" in env:
                    assert ps["degraded"] == 0 and stops == 0 and ps["pair_rounds_launches"] >= 1 and ps["crit_rounds_hops"] > 0, ps
                elif "MM_PAIR_PINJECT" in env:
                    assert ps["degraded"] == 1 and ps["pair_stops_inject"] >= 1 and ps["pair_stops_timeout"] == 0, ps
                elif "MM_PAIR_PTIMEOUT_US" in env:
                    assert ps["degraded"] == 1 and ps["pair_stops_timeout"] >= 1 and ps["pair_round_launches"] > 0, ps
                else:
                    assert ps["degraded"] == 1 and ps["pair_persist_off"] == 1 and ps["pair_rounds_launches"] == 0, ps


@pytest.mark.gpu
@pytest.mark.parametrize("ptiles", ["5", "6"])
def test_gpu_tile_length_never_grows_within_a_tick(gpu_cls, oracle_cls, monkeypatch, ptiles):
    """The pool tests/stress.py failed on with MM_PAIR_PTILES=5 in round 5 (and on round 4's sources): 260 000 players over
    all seven rating groups, +-60, 64 regions — sparse fits, so many next[] entries say NX_FAR.  With kp_rounds limited to
    five (six) tiles the two longest chains take turns in being too long for it; the batches in between were sized for
    forty tiles of 2048 positions, and kp_rounds then took over at 8192: NX_FAR entries computed for the short horizon
    were resolved from the end of the long one (tests/test_emu_tiled.py::test_the_tile_length_of_a_tick_never_grows has
    the mechanism).  46 lobbies of the 4000-5000 group were missing."""
    monkeypatch.setenv("MM_PAIR_PTILES", ptiles)
    rng = np.random.default_rng(130203984)
    n = 260000
    rating = rng.integers(0, 5001, size=n).astype(np.int32)
    cons = cons_make(np.zeros(n, np.int64), rng.integers(0, 64, size=n), 0, 0)
    cfg = make_config([mode_1v1(window=60, region_filter=True)], capacity=1 << 19)
    with gpu_cls(cfg) as a, oracle_cls(cfg) as b:
        assert np.array_equal(a.enqueue(rating, cons), b.enqueue(rating, cons))
        assert_same_tick(a.tick(0), b.tick(0), "tile length growth, MM_PAIR_PTILES=" + ptiles, SCORE_TOL)
        assert_same_state(a, b, cfg, "tile length growth")


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["1v1", "5v5"])
def test_gpu_two_engines_tick_concurrently(gpu_cls, oracle_cls, mode):
    """Two engines on one GPU, ticking at the same time from two host threads (own streams, own device memory): the 1v1
    pools are walked by kp_rounds launches whose workgroups wait for each other — two such launches compete for the same
    CUs, and one that finds them taken gives up and goes on launch by launch; the 5v5 pools by kt_fc launches whose
    chasers wait for kt_f's chunks.  Each engine against the oracle's run of its own pool, three rounds."""
    import threading
    if mode == "1v1":
        modes, kw, n = [mode_1v1(window=25, region_filter=True)], {}, 400000
    else:
        modes, kw, n = [mode_team(5, 2, 50, (1, 1, 1, 1, 1))], {"role_weights": ROLE_WEIGHTS_5V5}, 300000
    cfg = make_config(modes, capacity=1 << 19)
    pools = [make_pool(n, seed=31 + k, **kw) for k in range(2)]
    want = []
    for rating, cons in pools:
        with oracle_cls(cfg) as o:
            o.enqueue(rating, cons)
            want.append(o.tick(0))
    # (round 6) the two engines of one process run with DIFFERENT tuning — per-engine records (mm_engine_create_ex), where
    # rounds 1-5 read process-wide MM_* variables: the second one takes the short batches and the fall-back shapes
    tunings = [None, {"pair_ptiles": 8, "pair_pbatch": 6, "pair_batch": 5} if mode == "1v1" else
               {"team_f2": 3, "team_batch": 2, "team_late": 0, "team_rebuild": 3, "team_emit_max": 2}]
    engines = [gpu_cls(cfg, tunings[k]) for k in range(2)]
    assert engines[0].tuning() == gpu_cls.tuning_defaults()
    assert all(engines[1].tuning()[f] == v for f, v in tunings[1].items()) and engines[1].tuning() != engines[0].tuning()
    got = [[None] * 3 for _ in range(2)]
    errors = []
    start = threading.Barrier(2)

    def work(k):
        try:
            rating, cons = pools[k]
            for r in range(3):
                engines[k].reset()
                engines[k].enqueue(rating, cons)
                start.wait(timeout=120)
                got[k][r] = engines[k].tick(0)
        except Exception as ex:                       # never leave the other thread at the barrier
            errors.append(repr(ex))
            start.abort()

    th = [threading.Thread(target=work, args=(k,)) for k in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    for e in engines:
        e.close()
    assert not errors, errors
    for k in range(2):
        for r in range(3):
            assert_same_tick(got[k][r], want[k], "engine %d round %d" % (k, r), SCORE_TOL)


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["1v1", "5v5"])
def test_gpu_ticks_beside_a_foreign_workload(gpu_cls, oracle_cls, mode):
    """Somebody else's kernels on the same GPU while the engine ticks: a host thread keeps a torch stream busy with matrix
    products that fill every CU (each a few milliseconds), the engine ticks its pool three times beside them.  What waits for
    other workgroups inside a launch — kp_rounds' tiles for each other (1v1), kt_fc's chasers for kt_f's chunks and the
    emitters for their chasers (5v5) — finds its partners late or not on the chip; a stop or a flag that does not come is a
    slower tick, never a failed or a different one: every tick equals the oracle's."""
    import threading
    import torch
    if mode == "1v1":
        modes, kw, n = [mode_1v1(window=25, region_filter=True)], {}, 400000
    else:
        modes, kw, n = [mode_team(5, 2, 50, (1, 1, 1, 1, 1))], {"role_weights": ROLE_WEIGHTS_5V5}, 300000
    cfg = make_config(modes, capacity=1 << 19)
    rating, cons = make_pool(n, seed=41, **kw)
    with oracle_cls(cfg) as o:
        o.enqueue(rating, cons)
        want = o.tick(0)
    stop = threading.Event()
    launched = [0]

    def hog():
        s = torch.cuda.Stream()
        a = torch.randn(8192, 8192, device="cuda", dtype=torch.bfloat16)
        b = torch.randn(8192, 8192, device="cuda", dtype=torch.bfloat16)
        with torch.cuda.stream(s):
            while not stop.is_set():
                for _ in range(4):
                    a @ b
                    launched[0] += 1
                s.synchronize()

    th = threading.Thread(target=hog)
    th.start()
    try:
        with gpu_cls(cfg) as e:
            for r in range(3):
                e.reset()
                e.enqueue(rating, cons)
                got = e.tick(0)
                assert_same_tick(got, want, "beside a foreign workload, round %d" % r, SCORE_TOL)
    finally:
        stop.set()
        th.join(timeout=60)
    assert launched[0] > 0
