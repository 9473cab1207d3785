#!/usr/bin/env python
"""Where a step of bench.py's cfg-2 leg goes outside the walk: the binding's calls timed one by one on the host
(mm_reset, mm_enqueue_device, mm_tick, mm_matches), 30 steps of the seeded 1M-player 1v1 pool.
Usage (GPU box, repo root): python tools/step_breakdown.py [players]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from microservice_matchmaking_amd import Engine, make_config, mode_1v1  # noqa: E402
from microservice_matchmaking_amd.synth import make_pool  # noqa: E402
import ctypes as C  # noqa: E402
from microservice_matchmaking_amd._abi import MMStats, _ptr  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
rating, cons = make_pool(n, seed=1, dist="uniform", n_regions=8)
cfg = make_config([mode_1v1(window=25, region_filter=True)], capacity=1 << (n - 1).bit_length(), device=0, timing=True)
d_rating = torch.from_numpy(rating).cuda()
d_cons = torch.from_numpy(cons.view(np.int32)).cuda()
eng = Engine(cfg)
T = {"reset": [], "enqueue_device": [], "tick (C call)": [], "matches (C call)": [], "walk (events)": [], "step": []}
L = 2
slots = np.empty((n, L), np.uint32); score = np.empty(n, np.float32); group = np.empty(n, np.uint32); pass_ = np.empty(n, np.uint32)
for it in range(35):
    t0 = time.perf_counter()
    eng.reset()
    t1 = time.perf_counter()
    eng.enqueue_device(d_rating, d_cons)
    t2 = time.perf_counter()
    nm = C.c_uint32(); st = MMStats()
    eng._check(eng._fn("tick")(eng._h, 0, C.byref(nm), C.byref(st)), "tick")
    t3 = time.perf_counter()
    k = int(nm.value)
    eng._check(eng._fn("matches")(eng._h, 0, k, _ptr(slots[:k]), _ptr(score[:k]), _ptr(group[:k]), _ptr(pass_[:k])), "matches")
    t4 = time.perf_counter()
    if it >= 5:
        T["reset"].append(t1 - t0); T["enqueue_device"].append(t2 - t1); T["tick (C call)"].append(t3 - t2)
        T["matches (C call)"].append(t4 - t3); T["walk (events)"].append(st.as_dict()["walk_ms"] / 1e3); T["step"].append(t4 - t0)
for k, v in T.items():
    v = np.asarray(v) * 1e3
    print("%-18s mean %.3f ms  p50 %.3f  min %.3f  max %.3f" % (k, v.mean(), np.median(v), v.min(), v.max()))
eng.close()
