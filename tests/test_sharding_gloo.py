"""N > 1 path on CPU: 2, 4 and 8 ranks over gloo, the (mode, rating group) chains split between
them, no data-path collective.  The per-rank engine here is the oracle or the product's kernel
source under the CPU shim (the HIP engine needs a GPU); what is under test is the sharding
logic: routing, ownership (idle ranks included: 7 chains on 8 ranks), counter reduction, and that
the union of the ranks' emission lists is exactly the single-engine result — for a pool ticked
once and for the two-mode stream of BASELINE cfg-5."""
import os
import socket

import numpy as np
import pytest

from microservice_matchmaking_amd.config import make_config, mode_1v1, mode_team
from microservice_matchmaking_amd.sharding import (ChainSharding, GroupSharding, ShardedSearch, chain_weights,
                                                   rating_groups, tick_digests, union_digest)
from microservice_matchmaking_amd.stream import run_stream, stream_schedule
from microservice_matchmaking_amd.synth import ROLE_WEIGHTS_5V5, make_pool


def test_rating_groups_vectorised_matches_abi(oracle_cls):
    cfg = make_config([mode_1v1()])
    vals = [0, 1499, 1500, 1499.5, 5000, 5001, -1, 2999, 3000, float("nan"), 4000]
    with oracle_cls(cfg) as e:
        want = [e.find_rating_group(v) for v in vals]
    assert rating_groups(cfg, vals).tolist() == want


def test_lpt_assignment_balances():
    s = GroupSharding(7, 2, weights=[0.30, 0.10, 0.10, 0.10, 0.10, 0.10, 0.20])
    assert sorted(s.groups_of(0) + s.groups_of(1)) == list(range(7))
    assert abs(s.load[0] - s.load[1]) <= 0.1 + 1e-9
    s8 = GroupSharding(7, 8)
    assert len({int(o) for o in s8.owner}) == 7          # one group per rank, one rank idle
    assert len(s8.idle_ranks()) == 1


def test_chain_key_is_mode_and_group():
    """lobby_state.ex:74-83 selects the stored lobby by game mode: two modes of one rating group
    are separate chains and may live on different ranks."""
    w = np.array([[30, 10, 10, 10, 10, 10, 20], [13, 4, 4, 4, 4, 4, 9]], dtype=float)
    s = ChainSharding(2, 7, 8, w)
    assert s.idle_ranks() == []                          # 14 chains on 8 ranks: nobody idles
    assert sorted(c for r in range(8) for c in s.chains_of(r)) == [(m, g) for m in range(2) for g in range(7)]
    assert s.chain_owner[0, 0] != s.chain_owner[1, 0]    # the two heaviest chains of group 0 are apart
    assert s.load.max() == 30                            # bounded by the heaviest chain
    assert abs(s.bound() - w.sum() / 30) < 1e-9
    one = ChainSharding(1, 7, 8, w[0])
    assert one.idle_ranks() == [7] and abs(one.bound() - 100 / 30) < 1e-9


def test_chain_weights_counts_players():
    cfg = make_config([mode_1v1(), mode_team(5, 2, 50, (1, 1, 1, 1, 1))])
    rating, cons = make_pool(5000, seed=3, mode_weights=(70, 30), role_weights=ROLE_WEIGHTS_5V5)
    w = chain_weights(cfg, rating, cons)
    assert w.shape == (2, 7) and w.sum() == 5000
    grp = rating_groups(cfg, rating)
    assert w[1, 0] == int(((cons & 0xF) == 1)[grp == 0].sum())


def _case(kind, n):
    if kind == "1v1":
        return (make_config([mode_1v1(window=25, region_filter=True)], capacity=1 << 16),) + make_pool(n, seed=5)
    if kind == "5v5":
        return (make_config([mode_team(5, 2, 100, (1, 1, 1, 1, 1))], capacity=1 << 16),) + \
            make_pool(n, seed=6, role_weights=ROLE_WEIGHTS_5V5)
    if kind == "cfg4":                                   # BASELINE cfg-4's shape at a fifth of its size: the chains go through the tiled rounds
        return (make_config([mode_1v1(window=25, region_filter=True)], capacity=1 << 21),) + make_pool(n, seed=5)
    # two modes in one pool: 70 % 1v1 / 30 % 5v5 (BASELINE cfg-5's mix)
    cfg = make_config([mode_1v1(window=25, region_filter=True), mode_team(5, 2, 100, (1, 1, 1, 1, 1))],
                      capacity=1 << 16)
    rating, cons = make_pool(n, seed=7, mode_weights=(70, 30), role_weights=ROLE_WEIGHTS_5V5)
    cons = np.where((cons & 0xF) == 0, cons & ~np.uint32(0xF << 16), cons).astype(np.uint32)
    return cfg, rating, cons


def _engine_cls(engine):
    if engine == "oracle":
        from oracle.oracle import OracleEngine
        return OracleEngine
    if engine == "hip":                                  # the product library on cuda:0 (the gpu tier)
        from microservice_matchmaking_amd import Engine
        return Engine
    from emu_engine import EmuEngineSmall                # the product's kernel source under the CPU shim
    return EmuEngineSmall


def _worker(rank, world, port, n, out_q, kind="1v1", engine="oracle"):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    cfg, rating, cons = _case(kind, n)
    with ShardedSearch(cfg, _engine_cls(engine), rank, world, chain_weights(cfg, rating, cons)) as sh:
        idx, _ = sh.enqueue(rating, cons)
        digests, counters = {}, np.zeros(3)
        for md in range(cfg.n_modes):
            m = sh.tick(md)
            ids = sh.global_ids(m)
            mine = {c: d for c, d in tick_digests(md, cfg.n_groups, ids, m.group).items()
                    if sh.sharding.chain_owner[c] == rank}
            digests.update(mine)
            counters += [len(m), m.stats["pairs"], m.stats["pool_after"]]
        tot = ShardedSearch.sum_over_ranks(counters)
        gathered = [None] * world
        dist.all_gather_object(gathered, (digests, len(idx), sh.sharding.idle_ranks()))
        if rank == 0:
            out_q.put((tot, gathered))
    dist.barrier()
    dist.destroy_process_group()


def _spawn(target, world, args):
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=target, args=(r, world, port) + args[:1] + (q,) + args[1:]) for r in range(world)]
    for p in procs:
        p.start()
    out = q.get(timeout=400)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    return out


@pytest.mark.timeout(600)
@pytest.mark.parametrize("world,kind,engine,n", [
    (2, "1v1", "oracle", 20000), (2, "1v1", "emu", 12000), (2, "5v5", "emu", 8000),
    (4, "1v1", "oracle", 20000), (4, "mixed", "emu", 12000),
    (8, "1v1", "oracle", 20000),                          # 7 chains on 8 ranks: one rank idles
    (8, "mixed", "oracle", 30000),                        # 14 chains on 8 ranks: the (mode, group) key
])
def test_ranks_equal_one_engine(oracle_cls, world, kind, engine, n):
    """`emu`: every rank runs the product's kernel source (pair path / team path, small geometry)
    under the CPU shim on its share of the chains."""
    tot, gathered = _spawn(_worker, world, (n, kind, engine))
    cfg, rating, cons = _case(kind, n)
    want, counters = {}, np.zeros(3)
    with oracle_cls(cfg) as one:
        slots = one.enqueue(rating, cons)
        assert slots.tolist() == list(range(n))           # slot == global index on one engine
        for md in range(cfg.n_modes):
            ref = one.tick(md)
            want.update(tick_digests(md, cfg.n_groups, ref.slots.astype(np.int64), ref.group))
            counters += [len(ref), ref.stats["pairs"], ref.stats["pool_after"]]
    assert tot == counters.tolist()
    got = {}
    for digests, _, _ in gathered:
        assert not (set(digests) & set(got))              # every chain has exactly one owner
        got.update(digests)
    assert set(got) == set(want)
    assert union_digest(got) == union_digest(want)
    assert sum(g[1] for g in gathered) == n               # every player went to exactly one rank
    idle = gathered[0][2]
    assert len(idle) == max(0, world - cfg.n_modes * cfg.n_groups)
    for r in idle:
        assert gathered[r][1] == 0


STREAM = dict(qps=20000, seconds=0.6, tick_ms=10.0, seed=77)


def _stream_cfg():
    return make_config([mode_1v1(window=25, region_filter=True), mode_team(5, 2, 50, (1, 1, 1, 1, 1))],
                       capacity=1 << 16)


def _stream_worker(rank, world, port, engine, out_q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    cfg = _stream_cfg()
    # the expected share of every chain is known up front: mode mix x the uniform rating groups
    w = np.outer([0.7, 0.3], [0.30, 0.10, 0.10, 0.10, 0.10, 0.10, 0.20])
    with ShardedSearch(cfg, _engine_cls(engine or "oracle"), rank, world, w) as sh:
        res = run_stream(sh, stream_schedule(**STREAM), mode_weights=(70, 30), role_weights=ROLE_WEIGHTS_5V5,
                         realtime=False)
        mine = {c: d for c, d in res["digests"].items() if sh.sharding.chain_owner[c] == rank}
        floor = np.concatenate(res["floor"])
        gathered = [None] * world
        dist.all_gather_object(gathered, (mine, res["matched"], floor))
        if rank == 0:
            out_q.put(gathered)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
@pytest.mark.parametrize("world", [4, 8])
def test_two_mode_stream_sharded_by_chain(oracle_cls, world):
    """BASELINE cfg-5 on CPU: the 70/30 stream over `world` ranks, chains = (mode, group)."""
    gathered = _spawn(_stream_worker, world, (0,))
    cfg = _stream_cfg()
    with ShardedSearch(cfg, oracle_cls, 0, 1) as one:
        ref = run_stream(one, stream_schedule(**STREAM), mode_weights=(70, 30), role_weights=ROLE_WEIGHTS_5V5,
                         realtime=False)
    got = {}
    for mine, _, _ in gathered:
        got.update(mine)
    assert union_digest(got) == union_digest(ref["digests"])
    assert sum(g[1] for g in gathered) == ref["matched"] > 0
    # the arrival-limited latency floor is a property of the stream, not of the sharding
    a = np.sort(np.concatenate([g[2] for g in gathered]))
    b = np.sort(np.concatenate(ref["floor"]))
    assert np.array_equal(a, b)


@pytest.mark.gpu
@pytest.mark.timeout(900)
def test_gpu_two_ranks_two_real_engines_one_device(oracle_cls):
    """Round 6: the N > 1 path with the REAL library.  Two processes, rendezvous over gloo, each a
    ShardedSearch on a HIP engine — both on cuda:0, the only GPU of the box — a cfg-4-shaped 2M-player 1v1 pool split by
    chain (application.ex:26-40: rating groups never interact): the union of the two ranks' emission lists is the single
    oracle engine's, every player went to exactly one rank, the counters sum.  Until now the gloo tests ran the oracle or
    the shim engine, so two ranks had never met libmm_engine.so.  (RCCL itself cannot put two ranks on one device: the
    `nccl` branch of bench.py stays unexecuted until an N-GPU node runs it.)"""
    n = 2_000_000
    tot, gathered = _spawn(_worker, 2, (n, "cfg4", "hip"))
    cfg, rating, cons = _case("cfg4", n)
    with oracle_cls(cfg) as one:
        one.enqueue(rating, cons)
        ref = one.tick(0)
    want = tick_digests(0, cfg.n_groups, ref.slots.astype(np.int64), ref.group)
    assert tot == [len(ref), ref.stats["pairs"], ref.stats["pool_after"]]
    got = {}
    for digests, _, _ in gathered:
        assert not (set(digests) & set(got))
        got.update(digests)
    assert union_digest(got) == union_digest(want)
    assert sum(g[1] for g in gathered) == n and min(g[1] for g in gathered) > 0.3 * n


@pytest.mark.gpu
@pytest.mark.timeout(900)
def test_gpu_two_mode_stream_on_two_real_engines(oracle_cls):
    """... and BASELINE cfg-5's two-mode stream over the same two ranks: 14 chains = (mode, group) on two HIP engines."""
    gathered = _spawn(_stream_worker, 2, ("hip",))
    cfg = _stream_cfg()
    with ShardedSearch(cfg, oracle_cls, 0, 1) as one:
        ref = run_stream(one, stream_schedule(**STREAM), mode_weights=(70, 30), role_weights=ROLE_WEIGHTS_5V5,
                         realtime=False)
    got = {}
    for mine, _, _ in gathered:
        got.update(mine)
    assert union_digest(got) == union_digest(ref["digests"])
    assert sum(g[1] for g in gathered) == ref["matched"] > 0
