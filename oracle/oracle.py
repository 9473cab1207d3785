"""ctypes binding of the CPU oracle (oracle/libmm_oracle.so).  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

from microservice_matchmaking_amd._abi import EngineBase, MMConfig, MMStats, Matches, bind, _ptr

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "libmm_oracle.so")


def build():
    subprocess.check_call(["make", "-C", _HERE, "--no-print-directory"], stdout=subprocess.DEVNULL)


def load():
    if not os.path.exists(_LIB):
        build()
    lib = C.CDLL(_LIB)
    bind(lib, "mo_")
    lib.mo_tick_threads.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint32),
                                    C.POINTER(MMStats)]
    lib.mo_tick_threads.restype = C.c_int
    lib.mo_queue_slots.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint32),
                                   C.c_void_p]
    lib.mo_queue_slots.restype = C.c_int
    return lib


class OracleEngine(EngineBase):
    _prefix = "mo_"

    def __init__(self, cfg: MMConfig):
        if OracleEngine._lib is None:
            OracleEngine._lib = load()
        super().__init__(cfg)

    def tick_threads(self, mode=0, n_threads=7) -> Matches:
        n = C.c_uint32()
        st = MMStats()
        self._check(self._lib.mo_tick_threads(self._h, mode, n_threads, C.byref(n), C.byref(st)),
                    "tick_threads")
        n = int(n.value)
        L = self.lobby_size(mode)
        slots = np.empty((n, L), dtype=np.uint32)
        score = np.empty(n, dtype=np.float32)
        group = np.empty(n, dtype=np.uint32)
        pass_ = np.empty(n, dtype=np.uint32)
        self._check(self._lib.mo_matches(self._h, 0, n, _ptr(slots), _ptr(score), _ptr(group),
                                         _ptr(pass_)), "matches")
        return Matches(slots, score, group, pass_, st.as_dict())

    def queue_slots(self, mode, group):
        n = C.c_uint32(0)
        self._check(self._lib.mo_queue_slots(self._h, mode, group, C.byref(n), None), "queue_slots")
        out = np.zeros(max(1, n.value), dtype=np.uint32)
        n2 = C.c_uint32(out.shape[0])
        self._check(self._lib.mo_queue_slots(self._h, mode, group, C.byref(n2), _ptr(out)),
                    "queue_slots")
        return out[: n.value].copy()
