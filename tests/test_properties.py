"""Property tests (hypothesis) of the search path — SURVEY.md section 4 item 4: tied ratings in
every arrival order, all-same rating, zero feasible pairs, the cancelled-player mask.

Engines under test: the product's kernel source on the CPU shim (tests/emu, small geometry so that a
few hundred players walk tiles, routing and compaction) in the `not gpu` run, the HIP engine in the
`gpu` run (fewer examples).  The checker is the oracle; two properties also have closed forms that
need no oracle at all."""
import numpy as np
import pytest
from hypothesis import HealthCheck, given, settings, strategies as st

from emu_engine import EmuEngineSmall
from helpers import assert_same_state, assert_same_tick
from microservice_matchmaking_amd._abi import cons_make
from microservice_matchmaking_amd.config import make_config, mode_1v1, mode_team

CFG_1V1 = dict(modes=[mode_1v1(window=25, region_filter=True)], capacity=1 << 12)
CFG_2V2 = dict(modes=[mode_team(2, 2, 40, (1, 1))], capacity=1 << 12)

# few distinct ratings inside one rating group -> many exact ties; a handful of regions / roles
players = st.lists(st.tuples(st.sampled_from([1000, 1000, 1010, 1024, 1025, 1026, 1051, 1100, 1499]),
                             st.integers(0, 2), st.integers(0, 1)), min_size=0, max_size=400)


def arrays(ps):
    rating = np.asarray([p[0] for p in ps], dtype=np.int32)
    cons = cons_make(0, [p[1] for p in ps], 0, [p[2] for p in ps]) if ps else np.zeros(0, np.uint32)
    return rating, cons


def check_against_oracle(engine_cls, oracle_cls, cfg_kw, ps, cancel_idx=()):
    cfg = make_config(**cfg_kw)
    rating, cons = arrays(ps)
    if cfg.modes[0].n_roles == 1:
        cons = cons & ~np.uint32(0xF << 16)
    with engine_cls(cfg) as a, oracle_cls(cfg) as b:
        sa, sb = a.enqueue(rating, cons), b.enqueue(rating, cons)
        assert np.array_equal(sa, sb)
        if len(cancel_idx):
            cs = sa[np.asarray(sorted(set(cancel_idx)), dtype=np.int64)]
            a.cancel(cs)
            b.cancel(cs)
        for tick in range(2):                                  # the second tick starts from the stored lobby
            assert_same_tick(a.tick(0), b.tick(0), "tick %d" % tick)
            assert_same_state(a, b, cfg, "tick %d" % tick)


COMMON = dict(deadline=None, suppress_health_check=list(HealthCheck), derandomize=True)


@settings(max_examples=40, **COMMON)
@given(ps=players, seed=st.integers(0, 2 ** 16))
def test_ties_in_every_arrival_order_1v1(oracle_cls, ps, seed):
    """Tied ratings: first-fit must pick by queue position alone, whatever the arrival order."""
    ps = list(ps)
    np.random.default_rng(seed).shuffle(ps)
    check_against_oracle(EmuEngineSmall, oracle_cls, CFG_1V1, ps)


@settings(max_examples=25, **COMMON)
@given(ps=players)
def test_ties_team_mode(oracle_cls, ps):
    check_against_oracle(EmuEngineSmall, oracle_cls, CFG_2V2, ps)


@settings(max_examples=25, **COMMON)
@given(ps=players, data=st.data())
def test_cancelled_player_mask(oracle_cls, ps, data):
    """ActiveUser mask (active_user.ex:33-44): players cancelled before their first attempt leave no
    trace (docs/MATCH_CHECK.md section 3) — the tick equals the oracle's, and the lobbies are those of
    a pool the cancelled players never joined."""
    n = len(ps)
    cancel = data.draw(st.lists(st.integers(0, max(0, n - 1)), max_size=n // 3 if n else 0)) if n else []
    check_against_oracle(EmuEngineSmall, oracle_cls, CFG_1V1, ps, cancel)
    keep = [i for i in range(n) if i not in set(cancel)]
    cfg = make_config(**CFG_1V1)
    rating, cons = arrays(ps)
    cons = cons & ~np.uint32(0xF << 16)                        # 1v1 has one role
    with EmuEngineSmall(cfg) as masked, EmuEngineSmall(cfg) as never:
        sm = masked.enqueue(rating, cons)
        if cancel:
            masked.cancel(sm[np.asarray(sorted(set(cancel)), dtype=np.int64)])
        kidx = np.asarray(keep, dtype=np.int64)
        sn = never.enqueue(rating[kidx], cons[kidx])
        relabel = np.full(cfg.capacity, -1, dtype=np.int64)
        relabel[sn.astype(np.int64)] = sm[kidx]
        mm, mn = masked.tick(0), never.tick(0)
        assert mm.slots.shape == mn.slots.shape
        assert np.array_equal(mm.slots.astype(np.int64), relabel[mn.slots.astype(np.int64)])
        assert np.array_equal(mm.pass_, mn.pass_) and mm.stats["pairs"] == mn.stats["pairs"]


@settings(max_examples=15, **COMMON)
@given(n=st.integers(0, 700), rating=st.integers(0, 1499))
def test_all_same_rating_pairs_neighbours(n, rating):
    """Closed form, no oracle: with one rating and no filter every player fits everybody, so the
    cursor pairs the queue off in arrival order in ONE pass: (0,1), (2,3), ..."""
    cfg = make_config([mode_1v1(window=0)], capacity=1 << 12)
    with EmuEngineSmall(cfg) as e:
        s = e.enqueue(np.full(n, rating, np.int32), np.zeros(n, np.uint32))
        m = e.tick(0)
        assert m.slots.tolist() == [[int(s[2 * k]), int(s[2 * k + 1])] for k in range(n // 2)]
        assert not m.pass_.any() and not m.score.any()
        assert m.stats["pairs"] == n // 2                      # one evaluation per emitted lobby
        lob, _ = e.lobby_state(0, 0)
        assert lob.tolist() == ([int(s[-1])] if n % 2 else [])
        assert int(e.queue_depth(0).sum()) == 0


@settings(max_examples=15, **COMMON)
@given(n=st.integers(1, 14), window=st.integers(0, 40))
def test_zero_feasible_pairs_rotate_and_keep_their_order(n, window):
    """Closed form: ratings more than `window` apart -> nobody ever fits; the head seats as the anchor
    (starvation, MATCH_CHECK.md section 4), everybody else is rejected once per pass and re-enters at
    the tail (worker.ex:239-248 -> requeue/worker.ex:51-54), so the queue keeps its order."""
    cfg = make_config([mode_1v1(window=window)], capacity=1 << 12)
    rating = (np.arange(n, dtype=np.int32) * (window + 1))     # all inside [0, 1499]
    with EmuEngineSmall(cfg) as e:
        s = e.enqueue(rating, np.zeros(n, np.uint32))
        m = e.tick(0)
        assert len(m) == 0 and m.stats["pool_after"] == n
        assert e.lobby_state(0, 0)[0].tolist() == [int(s[0])]
        assert e.queue_slots(0, 0).tolist() == [int(x) for x in s[1:]]
        # pass 0: n-1 rejections against the anchor; pass 1: the same, seats nobody -> the tick ends
        assert m.stats["pairs"] == 2 * (n - 1) if n > 1 else m.stats["pairs"] == 0
        m2 = e.tick(0)                                          # a second tick changes nothing
        assert len(m2) == 0 and e.queue_slots(0, 0).tolist() == [int(x) for x in s[1:]]


# ---- the same properties on the HIP engine (short run; the driver's `-m gpu` tier) ----
@pytest.fixture(scope="module")
def gpu_cls():
    from microservice_matchmaking_amd import Engine
    return Engine


@pytest.mark.gpu
@settings(max_examples=12, **COMMON)
@given(ps=players, seed=st.integers(0, 2 ** 16), data=st.data())
def test_gpu_ties_and_cancel_mask(gpu_cls, oracle_cls, ps, seed, data):
    ps = list(ps)
    np.random.default_rng(seed).shuffle(ps)
    n = len(ps)
    cancel = data.draw(st.lists(st.integers(0, max(0, n - 1)), max_size=n // 4 if n else 0)) if n else []
    check_against_oracle(gpu_cls, oracle_cls, CFG_1V1, ps, cancel)
    check_against_oracle(gpu_cls, oracle_cls, CFG_2V2, ps)


@pytest.mark.gpu
def test_gpu_all_same_rating_and_zero_feasible_pairs(gpu_cls):
    cfg = make_config([mode_1v1(window=0)], capacity=1 << 16)
    for n in (0, 1, 2, 777, 40000):                            # 40000: the tiled pair path
        with gpu_cls(cfg) as e:
            s = e.enqueue(np.full(n, 1234, np.int32), np.zeros(n, np.uint32))
            m = e.tick(0)
            assert np.array_equal(m.slots, s[: 2 * (n // 2)].reshape(-1, 2)) and not m.pass_.any()
            assert e.lobby_state(0, 0)[0].tolist() == ([int(s[-1])] if n % 2 else [])
    cfg = make_config([mode_1v1(window=10)], capacity=1 << 16)
    with gpu_cls(cfg) as e:
        s = e.enqueue(np.arange(100, dtype=np.int32) * 11, np.zeros(100, np.uint32))
        m = e.tick(0)
        assert len(m) == 0 and e.queue_slots(0, 0).tolist() == s[1:].tolist()
        assert e.lobby_state(0, 0)[0].tolist() == [int(s[0])] and m.stats["pairs"] == 2 * 99
