"""Kernel control logic under the fiber shim (tests/emu): the same engine source that hipcc
builds for gfx950, compiled by g++ and run thread-by-thread on the CPU, compared with the
oracle.  This is not the parity gate (that is tests/test_gpu_parity.py on real hardware);
it catches logic bugs — tile boundaries, ballots, compaction, lobby handling — before a
GPU minute is spent."""
import numpy as np
import pytest

from emu_engine import EmuEngine
from helpers import assert_same_state, assert_same_tick, load_golden, random_scenario, run_golden_case
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
