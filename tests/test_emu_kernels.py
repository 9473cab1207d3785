"""Kernel control logic under the fiber shim (tests/emu): the same engine source that hipcc
builds for gfx950, compiled by g++ and run thread-by-thread on the CPU, compared with the
oracle.  This is not the parity gate (that is tests/test_gpu_parity.py on real hardware);
it catches logic bugs — tile boundaries, ballots, compaction, lobby handling — before a
GPU minute is spent."""
import numpy as np
import pytest

from emu_engine import EmuEngine
from helpers import assert_same_state, assert_same_tick, load_golden, random_scenario, run_golden_case
from microservice_matchmaking_amd._abi import cons_make
from microservice_matchmaking_amd.config import make_config, mode_1v1, mode_team
from microservice_matchmaking_amd.synth import ROLE_WEIGHTS_5V5, make_pool

GOLD = load_golden()


@pytest.mark.parametrize("case", GOLD["cases"], ids=[c["name"] for c in GOLD["cases"]])
def test_emu_golden_case(case):
    run_golden_case(EmuEngine, case)


MODE_SETS = {
    "1v1": [mode_1v1(window=60)],
    "1v1_region": [mode_1v1(window=40, region_filter=True)],
    "5v5_roles": [mode_team(5, 2, 400, (1, 1, 1, 1, 1))],
    "mixed": [mode_1v1(window=40, region_filter=True), mode_team(2, 2, 300, (1, 1)),
              mode_team(3, 2, 400, (3,), party_filter=True)],
    "3teams": [mode_team(2, 3, 500, (2,))],
}


@pytest.mark.parametrize("mset", sorted(MODE_SETS))
def test_emu_random_scenarios(oracle_cls, mset):
    cfg = make_config(MODE_SETS[mset], capacity=8192)
    rng = np.random.default_rng(7 + len(mset))
    with EmuEngine(cfg) as a, oracle_cls(cfg) as b:
        random_scenario(rng, cfg, a, b, n_rounds=3, batch=600, cancel_frac=0.05)


def test_emu_multi_tile_chain(oracle_cls):
    """A bronze-only pool longer than one 4096-element tile: anchors carried across tile and
    pass boundaries, in-place compaction over several tiles."""
    cfg = make_config([mode_1v1(window=25, region_filter=True)], capacity=16384)
    n = 9500
    rng = np.random.default_rng(5)
    rating = rng.integers(0, 1500, size=n).astype(np.int32)
    from microservice_matchmaking_amd._abi import cons_make
    cons = cons_make(0, rng.integers(0, 8, size=n), 0, 0)
    with EmuEngine(cfg) as a, oracle_cls(cfg) as b:
        assert np.array_equal(a.enqueue(rating, cons), b.enqueue(rating, cons))
        assert_same_tick(a.tick(0), b.tick(0), "multi-tile")
        assert_same_state(a, b, cfg)


def test_emu_synth_pool_5v5(oracle_cls):
    cfg = make_config([mode_team(5, 2, 200, (1, 1, 1, 1, 1))], capacity=8192)
    r, c = make_pool(3000, seed=3, dist="normal", role_weights=ROLE_WEIGHTS_5V5)
    with EmuEngine(cfg) as a, oracle_cls(cfg) as b:
        a.enqueue(r, c)
        b.enqueue(r, c)
        ma, mb = a.tick(0), b.tick(0)
        assert len(mb) > 20
        assert_same_tick(ma, mb, "5v5 synth")
        assert_same_state(a, b, cfg)


def test_emu_ring_wrap_and_full(oracle_cls):
    """Slot ring: wraps around capacity; refuses a batch that would overwrite live players."""
    from microservice_matchmaking_amd._abi import MMError, cons_make
    cfg = make_config([mode_1v1(window=5000)], capacity=8)
    for cls in (EmuEngine, oracle_cls):
        with cls(cfg) as e:
            s = e.enqueue(np.full(6, 1000, np.int32), cons_make(np.zeros(6)))
            assert s.tolist() == [0, 1, 2, 3, 4, 5]
            assert len(e.tick(0)) == 3                       # all six matched, slots free again
            s = e.enqueue(np.full(4, 1000, np.int32), cons_make(np.zeros(4)))
            assert s.tolist() == [6, 7, 0, 1]
            e.enqueue(np.asarray([4000], np.int32), cons_make(np.zeros(1)))   # slot 2, other group
            assert len(e.tick(0)) == 2
            e.enqueue(np.full(7, 1000, np.int32), cons_make(np.zeros(7)))     # slots 3..7,0,1
            with pytest.raises(MMError) as ei:
                e.enqueue(np.full(1, 1000, np.int32), cons_make(np.zeros(1)))  # slot 2 still live
            assert ei.value.status == -4


def test_emu_anchor_moves_to_a_lower_team_mid_lobby(oracle_cls):
    """A cancel empties team 1 of an open lobby; the next player seated there becomes the anchor
    (first player of the lowest-numbered non-empty team, MATCH_CHECK.md §2.1) in the middle of a
    64-candidate step, so the candidates after it are judged against the NEW anchor.  Found by
    tests/stress.py; pinned here against the oracle and the literal restatement."""
    from test_oracle_literal import literal_stage, literal_tick, to_payload
    cfg = make_config([mode_team(2, 2, 300, (1, 1))], capacity=256)
    from microservice_matchmaking_amd._abi import cons_make
    r1 = np.asarray([1000, 1200, 1250], np.int32)
    c1 = cons_make(0, 0, 0, [0, 0, 1])
    #        q0: no seat of role 0 in the stale lobby -> requeued;  q1 becomes the anchor (1450);
    #        q2 would fit the old anchor (1200) but not the new one;  q3, q4 complete the lobby
    r2 = np.asarray([950, 1450, 1000, 1300, 1500, 1210], np.int32)
    c2 = cons_make(0, 0, 0, [0, 1, 1, 0, 1, 0])
    stage = literal_stage(cfg)
    with EmuEngine(cfg) as a, oracle_cls(cfg) as b:
        for e in (a, b):
            assert e.enqueue(r1, c1).tolist() == [0, 1, 2]
        for s_, r_, c_ in zip([0, 1, 2], r1, c1):
            stage.deliver(to_payload(s_, r_, c_))
        assert_same_tick(a.tick(0), b.tick(0), "open the lobby")
        assert literal_tick(stage, cfg)[0] == []
        for e in (a, b):
            e.cancel(np.asarray([0, 2], np.uint32))          # team 1 of the lobby is gone
            assert e.enqueue(r2, c2).tolist() == [3, 4, 5, 6, 7, 8]
        for s_ in (0, 2):
            stage.cancel(s_)
        for s_, r_, c_ in zip(range(3, 9), r2, c2):
            stage.deliver(to_payload(s_, r_, c_))
        ma, mb = a.tick(0), b.tick(0)
        assert_same_tick(ma, mb, "anchor moved")
        assert_same_state(a, b, cfg)
        lit = literal_tick(stage, cfg)[0]
        assert [x[2] for x in lit] == mb.slots.tolist()
        assert 5 not in mb.slots.ravel().tolist()            # q2 (1000) never sat with anchor 1450


def test_emu_enqueue_steps_over_a_waiting_players_slot(oracle_cls):
    """A player nobody fits (alone in its rating group) keeps its slot while the ring wraps: later batches take the next FREE
    slots in ring order around it (the first version refused them with MM_ERR_FULL until the
    waiting player left — after capacity/qps seconds of a stream, for good)."""
    from microservice_matchmaking_amd._abi import MMError, cons_make
    cfg = make_config([mode_1v1(window=10)], capacity=8)
    for cls in (EmuEngine, oracle_cls):
        with cls(cfg) as e:
            s = e.enqueue(np.asarray([1000, 4000, 1000], np.int32), cons_make(np.zeros(3)))
            assert s.tolist() == [0, 1, 2]
            assert e.tick(0).slots.tolist() == [[0, 2]]                  # 4000 (slot 1) waits alone in its rating group
            for k, want in enumerate(([3, 4, 5, 6], [7, 0, 2, 3], [4, 5, 6, 7], [0, 2, 3, 4])):
                s = e.enqueue(np.full(4, 1000, np.int32), cons_make(np.zeros(4)))
                assert s.tolist() == want, (cls.__name__, k, s)          # slot 1 is stepped over every lap
                assert len(e.tick(0)) == 2
            assert int(e.queue_depth(0).sum()) + sum(len(e.lobby_state(0, g)[0]) for g in range(cfg.n_groups)) == 1
            # a player the mode rejects uses up its handle, as before
            s = e.enqueue(np.asarray([1000, 1000, 1000], np.int32), cons_make(np.asarray([0, 5, 0])))
            assert s.tolist() == [5, 0xFFFFFFFF, 7]
            s = e.enqueue(np.full(5, 2000, np.int32), cons_make(np.zeros(5)))   # 0,2,3,4,6: every free slot left
            assert s.tolist() == [0, 2, 3, 4, 6]
            with pytest.raises(MMError) as ei:
                e.enqueue(np.full(1, 1000, np.int32), cons_make(np.zeros(1)))
            assert ei.value.status == -4
            e.cancel(np.asarray([1], np.uint32))                          # the waiting player gives up
            assert len(e.tick(0)) == 3                                    # 5+7 and two pairs of the 2000s; slot 6 waits
            # slot 1 stays taken: a cancelled player seated in a lobby is only filtered out when a delivery
            # reaches its chain (MATCH_CHECK.md section 4, empty queue -> lobby untouched)
            assert e.enqueue(np.full(3, 3000, np.int32), cons_make(np.zeros(3))).tolist() == [7, 0, 2]


def test_emu_stream_laps_the_slot_ring_around_waiting_players(oracle_cls):
    from helpers import run_wrapping_stream
    laps, stepped = run_wrapping_stream(EmuEngine, oracle_cls, capacity=2048, ticks=60, per_tick=300)
    assert laps > 5 and stepped > 10, (laps, stepped)


def enqueue_device_rejects_leave_their_slots_free(engine_cls, to_device):
    """mm_enqueue_device with players the device refuses (mode not configured): only the accepted
    players hold a slot.  A refused player's slot is FREE — mm_cancel ignores it (no purge is armed
    for nothing) and the next batch that reaches it takes it."""
    import ctypes as C
    from microservice_matchmaking_amd._abi import MMEnqueueStats
    cfg = make_config([mode_1v1(window=50)], capacity=64)
    rating = np.full(40, 1000, np.int32) + np.arange(40, dtype=np.int32) * 100     # nobody fits anybody
    cons = cons_make(np.where(np.arange(40) % 4 == 3, 5, 0), 0, 0, 0)              # every fourth: mode 5
    with engine_cls(cfg) as e:
        fn = e._lib.mm_enqueue_device
        fn.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.POINTER(C.c_uint32), C.POINTER(MMEnqueueStats)]
        fn.restype = C.c_int
        d_r, d_c, keep = to_device(rating, cons)
        first, st = C.c_uint32(), MMEnqueueStats()
        assert fn(e._h, 40, d_r, d_c, C.byref(first), C.byref(st)) == 0
        assert (first.value, st.accepted, st.rejected) == (0, 30, 10)
        e.cancel(np.arange(3, 40, 4, dtype=np.uint32))        # the refused players' slots: nobody is there
        m = e.tick(0)
        assert len(m) == 0 and m.stats["pool_after"] == 30
        # 24 slots were never used + 10 slots of refused players = 34 free slots of 64
        s = e.enqueue(np.full(34, 4500, np.int32) + np.arange(34, dtype=np.int32), np.zeros(34, np.uint32))
        assert s.tolist() == list(range(40, 64)) + list(range(3, 40, 4))   # ring order from next_slot = 40
        with pytest.raises(Exception):
            e.enqueue(np.array([4600], np.int32), np.zeros(1, np.uint32))          # the pool is full now
        del keep


def test_emu_enqueue_device_rejects_leave_their_slots_free():
    def host_is_device(rating, cons):                         # under the shim device memory is host memory
        return rating.ctypes.data_as(C.c_void_p), cons.ctypes.data_as(C.c_void_p), (rating, cons)
    import ctypes as C
    enqueue_device_rejects_leave_their_slots_free(EmuEngine, host_is_device)


def test_matches_ranges_over_the_per_group_regions(oracle_cls):
    """mm_matches(first, count) for arbitrary ranges: the list of a tick lives in per-group regions of the pinned buffers
    (the lobbies leave for the host while the walk runs: a group's place is fixed by what it can emit at most), so a range
    may start and end inside any group's region; MM_ERR_RANGE past the end (include/mm_engine.h)."""
    import ctypes as C
    from emu_engine import EmuEngineSmall
    from microservice_matchmaking_amd._abi import MMError, _ptr
    cfg = make_config([mode_1v1(window=40, region_filter=True)], capacity=1 << 14)
    rating, cons = make_pool(9000, seed=17)
    with EmuEngineSmall(cfg) as e:
        e.enqueue(rating, cons)
        m = e.tick(0)
        n = len(m)
        assert n > 1000 and len(np.unique(m.group)) == 7
        rng = np.random.default_rng(1)
        edges = np.concatenate([[0, 1, n - 1, n], np.searchsorted(m.group, np.arange(1, 7)), rng.integers(0, n, 12)])
        for first in edges:
            for count in (0, 1, 2, 333, n - int(first)):
                count = int(min(count, n - int(first)))
                slots = np.full((max(count, 1), 2), 7, np.uint32)
                score = np.zeros(max(count, 1), np.float32)
                group = np.zeros(max(count, 1), np.uint32)
                pass_ = np.zeros(max(count, 1), np.uint32)
                rc = e._fn("matches")(e._h, int(first), count, _ptr(slots), _ptr(score), _ptr(group), _ptr(pass_))
                assert rc == 0
                a, b = int(first), int(first) + count
                assert np.array_equal(slots[:count], m.slots[a:b]) and np.array_equal(group[:count], m.group[a:b])
                assert np.array_equal(pass_[:count], m.pass_[a:b]) and np.array_equal(score[:count], m.score[a:b])
        assert e._fn("matches")(e._h, n, 1, None, None, None, None) == -8
        assert e._fn("matches")(e._h, 0, n + 1, None, None, None, None) == -8
        assert e._fn("matches")(e._h, n, 0, None, None, None, None) == 0


@pytest.mark.parametrize("what", ["pair", "team"])
def test_shim_randomised_stress_with_fuzzed_knobs(what, monkeypatch, capsys):
    """tests/stress.py --fuzz-knobs on the fiber-shim build (tiny geometry), twenty-five seconds each: every scenario's
    engine gets a random COMBINATION of mm_tuning fields through mm_engine_create_ex and every tick equals the oracle's.
    The CPU tier's share of round 6's lesson (the device's: test_gpu_randomised_stress_with_fuzzed_knobs; the soak:
    profiles/r06_stress_fuzz_knobs.txt)."""
    import importlib.util
    import os
    monkeypatch.setenv("MM_STRESS_ENGINE", "emu_small")
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "stress.py")
    spec = importlib.util.spec_from_file_location("shim_stress_fuzz_" + what, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    mod.main(["25", "11" if what == "pair" else "12"] + (["team"] if what == "team" else []) + ["--fuzz-knobs"])
    out = capsys.readouterr().out
    assert "--fuzz-knobs" in out and "scenarios ok" in out, out
