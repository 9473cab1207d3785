"""The pair (1v1) path's tiled kernels under the CPU shim, built with a tiny geometry
(tests/emu/libmm_engine_emu_small.so: PK_T=512, PL_MAX=1536) so that pools of a few thousand
players walk kp_tile_prep / kp_route / kp_tile_apply, compact, hand over to kp_late and come
back bit-exact against the oracle.  Logic tests only; the parity gate is test_gpu_parity.py."""
import numpy as np
import pytest

from emu_engine import EmuEngineSmall
from helpers import assert_same_state, assert_same_tick, random_scenario
from microservice_matchmaking_amd._abi import cons_make
from microservice_matchmaking_amd.config import make_config, mode_1v1, mode_team
from microservice_matchmaking_amd.synth import make_pool


def two_ticks(oracle_cls, n, seed, window, regions, lo=0, hi=5000):
    cfg = make_config([mode_1v1(window=window, region_filter=regions > 1)], capacity=32768)
    rng = np.random.default_rng(seed)
    with EmuEngineSmall(cfg) as a, oracle_cls(cfg) as b:
        for k in range(2):
            nn = n if k == 0 else n // 2
            rating = rng.integers(lo, hi + 1, size=nn).astype(np.int32)
            cons = cons_make(0, rng.integers(0, regions, size=nn), 0, 0)
            assert np.array_equal(a.enqueue(rating, cons), b.enqueue(rating, cons))
            ma, mb = a.tick(0), b.tick(0)
            assert_same_tick(ma, mb, "tick %d" % k)
            assert_same_state(a, b, cfg, "tick %d" % k)
        return len(ma)


def test_tiled_sparse_pool_far_and_none(oracle_cls):
    """+-2 window: almost nobody fits; exit anchors without a partner inside the horizon."""
    two_ticks(oracle_cls, 6000, seed=1, window=2, regions=4)


def test_tiled_dense_pool(oracle_cls):
    """+-500 window: nearly every player's partner is its neighbour (long in-tile chains)."""
    assert two_ticks(oracle_cls, 8000, seed=2, window=500, regions=4) > 500


def test_tiled_typical_pool_compacts_and_hands_over(oracle_cls):
    assert two_ticks(oracle_cls, 7000, seed=7, window=60, regions=4) > 100


def test_tiled_single_group_long_chain(oracle_cls):
    """Everybody in one rating group: one chain of ten tiles."""
    two_ticks(oracle_cls, 5000, seed=3, window=30, regions=2, lo=0, hi=1400)


def test_tiled_with_cancels_and_mode_mix(oracle_cls):
    """Cancel ticks take the generic kernel, the others the pair path; lobbies, queues and the
    anchor (also one left in team 2 by a cancel) carry over between the two."""
    cfg = make_config([mode_1v1(window=80, region_filter=True), mode_team(2, 2, 300, (1, 1))],
                      capacity=16384)
    rng = np.random.default_rng(11)
    with EmuEngineSmall(cfg) as a, oracle_cls(cfg) as b:
        random_scenario(rng, cfg, a, b, n_rounds=4, batch=3000, cancel_frac=0.03)


def test_tiled_extreme_ratings_fall_back(oracle_cls):
    """A chain whose rating span does not fit the packed key is walked by the generic kernel."""
    cfg = make_config([mode_1v1(window=50)], capacity=8192)
    rng = np.random.default_rng(5)
    rating = rng.integers(3000, 3400, size=3000).astype(np.int32)
    rating[::500] = np.asarray([2**31 - 1, -2**31, 2**31 - 5, -2**31 + 7, 2**30, -2**30], np.int32)
    cons = cons_make(np.zeros(3000))
    grp = np.full(3000, 4, np.uint8)          # host routed them all to one group
    with EmuEngineSmall(cfg) as a, oracle_cls(cfg) as b:
        assert np.array_equal(a.enqueue(rating, cons, grp), b.enqueue(rating, cons, grp))
        assert_same_tick(a.tick(0), b.tick(0), "extreme")
        assert_same_state(a, b, cfg)


def test_forced_generic_walk(oracle_cls, monkeypatch):
    """MM_FORCE_GENERIC=1 walks 1v1 modes with k_walk as well (the A/B switch of the bench)."""
    monkeypatch.setenv("MM_FORCE_GENERIC", "1")
    two_ticks(oracle_cls, 3000, seed=10, window=40, regions=4)


@pytest.mark.parametrize("seed", [0, 1, 2, 3])
def test_tiled_cancel_ticks_on_the_pair_path(oracle_cls, seed):
    """Cancel ticks are walked by the pair path too: the open lobby's anchor, the head of the queue
    and queued players are cancelled between ticks (stale-lobby rule of MATCH_CHECK.md §4), on
    chains long enough for the tiled rounds."""
    rng = np.random.default_rng(seed)
    window = int(rng.choice([1, 5, 40]))
    cfg = make_config([mode_1v1(window=window, region_filter=True)], capacity=16384)
    with EmuEngineSmall(cfg) as a, oracle_cls(cfg) as b:
        for k in range(4):
            n = int(rng.choice([0, 1, 2, 300, 2500, 4000]))
            rating = rng.integers(0, 1400, size=n).astype(np.int32)
            cons = cons_make(0, rng.integers(0, 3, size=n), 0, 0)
            assert np.array_equal(a.enqueue(rating, cons), b.enqueue(rating, cons))
            cs = []
            ls, _ = b.lobby_state(0, 0)
            if len(ls) and rng.integers(0, 2):
                cs += ls.tolist()
            qs = b.queue_slots(0, 0)
            if len(qs) and rng.integers(0, 2):
                cs.append(int(qs[0]))
            if len(qs) > 5:
                cs += rng.choice(qs, size=3, replace=False).tolist()
            if cs:
                cs = np.unique(np.asarray(cs, np.uint32))
                a.cancel(cs)
                b.cancel(cs)
            assert_same_tick(a.tick(0), b.tick(0), "seed %d tick %d" % (seed, k))
            assert_same_state(a, b, cfg)


def test_product_geometry_walks_every_tile_length(oracle_cls, monkeypatch):
    """The product geometry under the shim (tiles of 8192 / 4096 / 2048 positions chosen batch by
    batch as the chain shrinks, then the LDS-resident kernel below 16384 players): one 1v1 pool whose
    only rating group starts at 50k players, against the oracle.  MM_PAIR_TILES=10 (the cap on the tiles of
    the longest chain, 40 in production) makes a pool of this size pass through all three lengths."""
    from emu_engine import EmuEngine
    monkeypatch.setenv("MM_PAIR_TILES", "10")
    cfg = make_config([mode_1v1(window=25, region_filter=True)], capacity=1 << 16)
    rating, cons = make_pool(50000, seed=9)
    rating = (rating % 1400).astype(np.int32)                 # everybody in the first rating group
    with EmuEngine(cfg) as e, oracle_cls(cfg) as o:
        assert np.array_equal(e.enqueue(rating, cons), o.enqueue(rating, cons))
        assert_same_tick(e.tick(0), o.tick(0), "50k players, one chain")
        assert_same_state(e, o, cfg, "50k players, one chain")


@pytest.mark.parametrize("env", [{"MM_PAIR_PERSIST": "0"}, {"MM_PAIR_PINJECT": "2"}, {"MM_PAIR_PINJECT": "1"}, {"MM_PAIR_PBATCH": "5"},
                                 {"MM_PAIR_PTILES": "3"}],
                         ids=lambda e: ",".join("%s=%s" % kv for kv in sorted(e.items())))
def test_tiled_rounds_stop_and_go_on(oracle_cls, monkeypatch, env):
    """kp_rounds' ways out — the batch ends, the longest chain yields for its compaction, a workgroup declares a stop
    (MM_PAIR_PINJECT) — and the one-launch-per-pass path it falls back to (MM_PAIR_PERSIST=0: kp_round alone), on the
    shim: every one of them leaves the chain committed, the results are the oracle's."""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    two_ticks(oracle_cls, 7000, seed=21, window=40, regions=3)
    if "MM_PAIR_PINJECT" in env:                          # one long chain as well: ten tiles, the stop in mid-batch
        two_ticks(oracle_cls, 4000, seed=22, window=30, regions=2, lo=0, hi=1400)


@pytest.mark.parametrize("env,want", [({}, "clean"), ({"MM_PAIR_PINJECT": "2"}, "inject"), ({"MM_PAIR_PERSIST": "0"}, "off")],
                         ids=["default", "MM_PAIR_PINJECT=2", "MM_PAIR_PERSIST=0"])
def test_path_stats_say_which_launch_shapes_a_tick_took(oracle_cls, monkeypatch, env, want):
    """mm_path_stats_get (include/mm_engine.h): a fall-back of the persistent launch used to leave one MM_PAIR_DEBUG line on
    stderr and `ok: true` everywhere ("What's weak" 8).  Now the tick's record says how many kp_rounds /
    kp_round launches it took, which stops it met — each counted ONCE, at the look behind the launch that declared it
    (PairChain.pfail stays in the record and used to be counted again at every later look of the tick) — and
    `degraded` when a fall-back was in force.  Results are the oracle's either way."""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    cfg = make_config([mode_1v1(window=30, region_filter=True)], capacity=16384)
    rng = np.random.default_rng(31)
    rating = rng.integers(0, 1401, size=5000).astype(np.int32)        # one chain of ten tiles (PK_T = 512 in this build)
    cons = cons_make(0, rng.integers(0, 2, size=5000), 0, 0)
    with EmuEngineSmall(cfg) as a, oracle_cls(cfg) as b:
        assert a.path_stats()["mode"] == 0xFFFFFFFF and b.path_stats() is None
        assert np.array_equal(a.enqueue(rating, cons), b.enqueue(rating, cons))
        ma, mb = a.tick(0), b.tick(0)
        assert_same_tick(ma, mb, "path stats")
        ps = a.path_stats()
    assert ps["mode"] == 0 and ps["paths"] == 2 and ps["host_looks"] >= 2
    assert ps["crit_group"] == 0 and ps["crit_passes"] == ma.stats["passes_max"]
    assert ps["crit_rounds_passes"] + ps["crit_round_passes"] + ps["crit_late_passes"] == ps["crit_passes"]
    assert ps["crit_late_lobbies"] <= len(ma) and ps["pair_tiled_passes"] == ps["crit_rounds_passes"] + ps["crit_round_passes"]
    stops = ps["pair_stops_timeout"] + ps["pair_stops_xcd"] + ps["pair_stops_inject"]
    assert stops == ps["pair_stops_total"]                            # a fresh engine: every stop of its life was this tick's, once
    if want == "clean":
        assert ps["degraded"] == 0 and stops == 0 and ps["pair_rounds_launches"] >= 1 and ps["pair_persist_off"] == 0
        assert ps["crit_rounds_passes"] == ps["pair_rounds_passes"] > 0 and ps["crit_rounds_hops"] > 0
    elif want == "inject":
        assert ps["degraded"] == 1 and ps["pair_stops_inject"] == stops >= 1 and ps["pair_round_launches"] > 0
        assert stops <= ps["pair_rounds_launches"]                    # one chain: at most one stop per launch
    else:
        assert ps["degraded"] == 1 and ps["pair_persist_off"] == 1 and ps["pair_rounds_launches"] == 0 and stops == 0
        assert ps["crit_rounds_passes"] == 0 and ps["pair_round_launches"] > 0


@pytest.mark.parametrize("seed", [18, 21, 39])
def test_the_tile_length_of_a_tick_never_grows(oracle_cls, monkeypatch, seed):
    """Two chains of similar length, sparse fits, kp_rounds limited to four tiles (MM_PAIR_PTILES=4).  Both start too long
    for kp_rounds and are walked launch by launch at the SHORT tile length (the cap of a kp_round batch was ten tiles
    here, forty in the product); the longer one is compacted when it fits, the other keeps the batches launch by launch
    for a while — and the first one's fresh next[] entries say NX_FAR, "nobody inside the horizon of two SHORT tiles";
    when the second fits as well, kp_rounds took over at the LONG tile length (its own cap), where the walk resolves
    NX_FAR from the end of two long tiles: the stretch in between was never looked at — a later partner, or none.
    Found in round 5 by tests/stress.py with MM_PAIR_PTILES=5 on the device (seed 130203984: 46 lobbies of one chain
    missing; round 4's sources did the same); in the product the same growth needs a kp_rounds stop and its cool-down.
    One cap for both kinds of batch now, and a guard that compacts every tiled chain should a tile length grow all the
    same.  These three seeds differed from the oracle on round 4's build of this geometry."""
    monkeypatch.setenv("MM_PAIR_PTILES", "4")
    rng = np.random.default_rng(seed)
    ra, rb = rng.integers(4000, 5001, size=2560), rng.integers(0, 1500, size=2500)
    rating = np.concatenate([ra, rb]).astype(np.int32)
    rating = rating[rng.permutation(rating.size)]
    cons = cons_make(np.zeros(rating.size, np.int64), rng.integers(0, 16, size=rating.size), 0, 0)
    cfg = make_config([mode_1v1(window=25, region_filter=True)], capacity=1 << 14)
    with EmuEngineSmall(cfg) as a, oracle_cls(cfg) as b:
        assert np.array_equal(a.enqueue(rating, cons), b.enqueue(rating, cons))
        assert_same_tick(a.tick(0), b.tick(0), "tile length growth, seed %d" % seed)
        assert_same_state(a, b, cfg, "tile length growth, seed %d" % seed)
        ps = a.path_stats()
        assert ps["pair_round_launches"] > 0 and ps["pair_rounds_launches"] > 0      # both kinds of batch were in the tick
