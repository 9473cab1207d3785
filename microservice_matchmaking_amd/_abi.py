"""ctypes mirror of include/mm_engine.h and a thin object wrapper over it.

`bind(lib, prefix)` types every export; `EngineBase` drives any library that exports the
ABI under a prefix.  The product (`engine.Engine`) binds libmm_engine.so with prefix
``mm_``; the test-only oracle binds its own library with ``mo_`` — the two are driven by
identical calls, which is what makes the parity tests read like one test run twice.
"""
from __future__ import annotations

import ctypes as C
import time
from dataclasses import dataclass

import numpy as np

MM_ABI_VERSION = 1
MM_MAX_GROUPS = 16
MM_MAX_MODES = 16
MM_MAX_ROLES = 8
MM_MAX_TEAMS = 4
MM_MAX_LOBBY = 16
MM_MODE_REGION_FILTER = 1
MM_MODE_PARTY_FILTER = 2
MM_CFG_TIMING = 1
NO_SLOT = 0xFFFFFFFF

STATUS_NAMES = {
    0: "MM_OK", -1: "MM_ERR_INVALID_ARG", -2: "MM_ERR_NO_DEVICE", -3: "MM_ERR_OOM",
    -4: "MM_ERR_FULL", -5: "MM_ERR_HIP", -6: "MM_ERR_INTERNAL", -7: "MM_ERR_ABI",
    -8: "MM_ERR_RANGE", -9: "MM_ERR_STATE",
}


class MMRatingGroup(C.Structure):
    _fields_ = [("from_", C.c_int32), ("to", C.c_int32)]


class MMModeConfig(C.Structure):
    _fields_ = [
        ("team_size", C.c_uint32), ("teams", C.c_uint32), ("window", C.c_uint32),
        ("flags", C.c_uint32), ("n_roles", C.c_uint32),
        ("role_quota", C.c_uint8 * MM_MAX_ROLES),
    ]


class MMConfig(C.Structure):
    _fields_ = [
        ("abi_version", C.c_uint32), ("n_groups", C.c_uint32),
        ("groups", MMRatingGroup * MM_MAX_GROUPS), ("default_group", C.c_uint32),
        ("n_modes", C.c_uint32), ("modes", MMModeConfig * MM_MAX_MODES),
        ("capacity", C.c_uint32), ("device", C.c_int32), ("flags", C.c_uint32),
    ]


class MMStats(C.Structure):
    _fields_ = [
        ("pool_before", C.c_uint32), ("pool_after", C.c_uint32), ("matches", C.c_uint32),
        ("players_matched", C.c_uint32), ("passes_max", C.c_uint32), ("chains", C.c_uint32),
        ("pairs", C.c_uint64), ("scanned", C.c_uint64),
        ("walk_ms", C.c_float), ("filter_ms", C.c_float), ("copy_ms", C.c_float),
        ("total_ms", C.c_float),
    ]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


class MMEnqueueStats(C.Structure):
    _fields_ = [("accepted", C.c_uint32), ("rejected", C.c_uint32),
                ("bucket_ms", C.c_float), ("total_ms", C.c_float)]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


class MMPathStats(C.Structure):
    """include/mm_engine.h mm_path_stats: the launch shapes and fall-backs of the last tick (size-versioned)."""
    _fields_ = [(n, C.c_uint32) for n in (
        "size", "mode", "paths", "host_looks",
        "pair_rounds_launches", "pair_rounds_passes", "pair_round_launches", "pair_tiled_passes",
        "pair_stops_timeout", "pair_stops_xcd", "pair_stops_inject", "pair_yields",
        "pair_persist_off", "pair_cooldown", "pair_stops_total",
        "team_f_launches", "team_fc_launches", "team_late_launches", "team_build_launches",
        "team_flags_late", "team_flags_late_total",
        "crit_group", "crit_passes", "crit_rounds_passes", "crit_rounds_hops", "crit_round_passes", "crit_late_passes",
        "crit_late_lobbies", "degraded",
        "crit_timed_passes", "crit_timed_hops", "crit_barrier_cycles", "crit_hop_cycles", "clk_cycles", "clk_wall_ticks",
        "pair_nx_init_ns", "pair_tested_lo", "pair_tested_hi", "pair_tested_nx_lo", "pair_tested_nx_hi",
        "crit_team_group", "crit_team_passes", "crit_team_f_passes", "crit_team_fc_passes", "crit_team_late_passes",
        "crit_team_f_lobbies", "crit_team_fc_lobbies", "crit_team_late_lobbies", "crit_team_lookups", "crit_team_late_lookups")]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_ if n != "size"}


class MMError(RuntimeError):
    def __init__(self, status, where):
        self.status = int(status)
        super().__init__("%s failed: %s (%d)" % (where, STATUS_NAMES.get(self.status, "?"), self.status))


def cons_make(mode=0, region=0, party=0, role=0):
    """MM_CONS_MAKE for scalars or numpy arrays."""
    mode, region, party, role = (np.asarray(x, dtype=np.uint32) for x in (mode, region, party, role))
    return ((mode & 0xF) | ((region & 0xFF) << 4) | ((party & 0xF) << 12) | ((role & 0xF) << 16)).astype(np.uint32)


# Names every ABI library must export (suffixes after the prefix); tests/test_abi_symbols.py
# cross-checks this list against include/mm_engine.h.
ABI_FUNCTIONS = [
    "engine_create", "engine_destroy", "reset", "find_rating_group", "enqueue", "cancel",
    "tick", "matches", "queue_depth", "queue_slots", "lobby_state",
]
PRODUCT_ONLY_FUNCTIONS = ["abi_version", "strerror", "config_default", "enqueue_device",
                          "last_hip_error", "path_stats_get", "engine_create_ex", "tuning_default", "tuning_set",
                          "tuning_name", "tuning_get"]


class MMCodecCfg(C.Structure):
    """include/mm_codec.h mm_codec_cfg."""
    _fields_ = [("n_modes", C.c_uint32), ("mode_name", C.c_char_p * 16),
                ("region_key", C.c_char_p), ("party_key", C.c_char_p), ("role_key", C.c_char_p)]


DEC_OK, DEC_BAD_JSON, DEC_NO_MODE, DEC_BAD_FIELD, DEC_RATING_INEXACT, DEC_RATING_NOT_NUMBER = range(6)


def decode_players(lib, cfg, mode_names, messages, region_key=None, party_key=None, role_key=None):
    """mm_decode_players over a list of `bytes` payloads -> dict of numpy columns."""
    cc = MMCodecCfg()
    cc.n_modes = len(mode_names)
    for k, name in enumerate(mode_names):
        cc.mode_name[k] = name.encode()
    cc.region_key = region_key.encode() if region_key else None
    cc.party_key = party_key.encode() if party_key else None
    cc.role_key = role_key.encode() if role_key else None
    n = len(messages)
    off = np.zeros(n + 1, dtype=np.uint64)
    if n:
        off[1:] = np.cumsum([len(m) for m in messages])
    buf = b"".join(messages)
    out = {"rating": np.zeros(n, np.int32), "cons": np.zeros(n, np.uint32), "group": np.zeros(n, np.uint8),
           "status": np.zeros(n, np.uint8), "id_off": np.zeros(n, np.uint32), "id_len": np.zeros(n, np.uint32)}
    fn = lib.mm_decode_players
    fn.restype = C.c_int
    fn.argtypes = [C.POINTER(MMConfig), C.POINTER(MMCodecCfg), C.c_char_p, C.c_void_p, C.c_uint32] + [C.c_void_p] * 6
    t0 = time.perf_counter()
    rc = fn(C.byref(cfg), C.byref(cc), buf, _ptr(off), n, *[_ptr(out[k]) for k in
                                                            ("rating", "cons", "group", "status", "id_off", "id_len")])
    seconds = time.perf_counter() - t0
    if rc != 0:
        raise MMError(rc, "mm_decode_players")
    out["seconds"] = seconds                  # the C call alone (batch assembly above is Python's)
    out["ids"] = [messages[i][int(out["id_off"][i]):int(out["id_off"][i]) + int(out["id_len"][i])] for i in range(n)]
    return out


def encode_lobby(lib, game_mode, teams, team_size, payloads):
    """mm_encode_lobby: the L payloads of one lobby (team major) -> the published JSON (bytes)."""
    n = len(payloads)
    arr = (C.c_char_p * n)(*payloads)
    lens = np.asarray([len(p) for p in payloads], dtype=np.uint32)
    fn = lib.mm_encode_lobby
    fn.restype = C.c_int
    fn.argtypes = [C.c_char_p, C.c_uint32, C.c_uint32, C.POINTER(C.c_char_p), C.c_void_p, C.c_void_p, C.c_uint64,
                   C.POINTER(C.c_uint64)]
    w = C.c_uint64()
    rc = fn(game_mode.encode("utf-8"), teams, team_size, arr, _ptr(lens), None, 0, C.byref(w))
    if rc not in (0, -8):                     # MM_ERR_RANGE: the size query
        raise MMError(rc, "mm_encode_lobby")
    buf = np.empty(int(w.value), dtype=np.uint8)
    rc = fn(game_mode.encode("utf-8"), teams, team_size, arr, _ptr(lens), _ptr(buf), C.c_uint64(buf.size), C.byref(w))
    if rc != 0:
        raise MMError(rc, "mm_encode_lobby")
    return buf.tobytes()


def bind(lib, prefix):
    """Set argtypes/restype for the common ABI under `prefix` ('mm_' or 'mo_')."""
    u32p = C.POINTER(C.c_uint32)
    f = lambda n: getattr(lib, prefix + n)
    f("engine_create").argtypes = [C.POINTER(MMConfig), C.POINTER(C.c_void_p)]
    f("engine_create").restype = C.c_int
    f("engine_destroy").argtypes = [C.c_void_p]
    f("engine_destroy").restype = None
    f("reset").argtypes = [C.c_void_p]
    f("reset").restype = C.c_int
    f("find_rating_group").argtypes = [C.POINTER(MMConfig), C.c_double, u32p]
    f("find_rating_group").restype = C.c_int
    f("enqueue").argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p,
                             C.c_void_p, C.POINTER(MMEnqueueStats)]
    f("enqueue").restype = C.c_int
    f("cancel").argtypes = [C.c_void_p, C.c_uint32, C.c_void_p]
    f("cancel").restype = C.c_int
    f("tick").argtypes = [C.c_void_p, C.c_uint32, u32p, C.POINTER(MMStats)]
    f("tick").restype = C.c_int
    f("matches").argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p,
                             C.c_void_p, C.c_void_p]
    f("matches").restype = C.c_int
    f("queue_depth").argtypes = [C.c_void_p, C.c_uint32, C.c_void_p]
    f("queue_depth").restype = C.c_int
    f("queue_slots").argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, u32p, C.c_void_p]
    f("queue_slots").restype = C.c_int
    f("lobby_state").argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, u32p, C.c_void_p, C.c_void_p]
    f("lobby_state").restype = C.c_int
    if hasattr(lib, prefix + "snapshot"):                 # the oracle does not mirror these
        u64p = C.POINTER(C.c_uint64)
        f("snapshot_size").argtypes = [C.c_void_p, u64p]
        f("snapshot_size").restype = C.c_int
        f("snapshot").argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, u64p]
        f("snapshot").restype = C.c_int
        f("restore").argtypes = [C.c_void_p, C.c_void_p, C.c_uint64]
        f("restore").restype = C.c_int
    return lib


@dataclass
class Matches:
    """Result of one tick, in emission order."""
    slots: np.ndarray   # (n, L) uint32, team order
    score: np.ndarray   # (n,) float32
    group: np.ndarray   # (n,) uint32
    pass_: np.ndarray   # (n,) uint32
    stats: dict

    def __len__(self):
        return int(self.slots.shape[0])


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


class EngineBase:
    """Object wrapper over one ABI library.  Subclasses set `_lib` and `_prefix`."""

    _lib = None
    _prefix = "mm_"

    def __init__(self, cfg: MMConfig, tuning=None):
        """tuning: {field of include/mm_engine.h's mm_tuning: value} for THIS engine (mm_engine_create_ex); fields
        not named keep their defaults (built-in, or the MM_* environment variable's).  Results never depend on it."""
        self.cfg = cfg
        self._h = C.c_void_p()
        if tuning:
            rec = self.tuning_record(tuning)
            fn = self._fn("engine_create_ex")
            fn.argtypes = [C.POINTER(MMConfig), C.c_void_p, C.POINTER(C.c_void_p)]
            fn.restype = C.c_int
            self._check(fn(C.byref(cfg), rec, C.byref(self._h)), "engine_create_ex")
        else:
            self._check(self._fn("engine_create")(C.byref(cfg), C.byref(self._h)), "engine_create")

    # mm_tuning: addressed by NAME through the library (no mirror of the layout here) --------------
    TUNING_WORDS = 128                        # room for the record of any library version (it fills what it knows)

    @classmethod
    def tuning_record(cls, tuning=None):
        """A buffer holding mm_tuning_default() with `tuning` laid over it by mm_tuning_set."""
        lib, pre = cls._lib, cls._prefix
        rec = (C.c_uint32 * cls.TUNING_WORDS)()
        rec[0] = 4 * cls.TUNING_WORDS
        dflt, setf = getattr(lib, pre + "tuning_default"), getattr(lib, pre + "tuning_set")
        dflt.argtypes, dflt.restype = [C.c_void_p], C.c_int
        setf.argtypes, setf.restype = [C.c_void_p, C.c_char_p, C.c_uint32], C.c_int
        rc = dflt(rec)
        if rc != 0:
            raise MMError(rc, pre + "tuning_default")
        for k, v in (tuning or {}).items():
            rc = setf(rec, k.encode(), int(v))
            if rc != 0:
                raise MMError(rc, "%stuning_set(%s=%r)" % (pre, k, v))
        return rec

    @classmethod
    def tuning_names(cls):
        """The fields of mm_tuning, in order (mm_tuning_name)."""
        fn = getattr(cls._lib, cls._prefix + "tuning_name")
        fn.argtypes, fn.restype = [C.c_uint32], C.c_char_p
        out = []
        while True:
            nm = fn(len(out))
            if nm is None:
                return out
            out.append(nm.decode())

    @classmethod
    def tuning_defaults(cls):
        rec = cls.tuning_record()
        return {n: int(rec[1 + i]) for i, n in enumerate(cls.tuning_names())}

    def tuning(self):
        """mm_tuning_get: what this engine runs with."""
        fn = self._fn("tuning_get")
        fn.argtypes, fn.restype = [C.c_void_p, C.c_void_p], C.c_int
        rec = (C.c_uint32 * self.TUNING_WORDS)()
        rec[0] = 4 * self.TUNING_WORDS
        self._check(fn(self._h, rec), "tuning_get")
        return {n: int(rec[1 + i]) for i, n in enumerate(self.tuning_names())}

    # plumbing ------------------------------------------------------------------------
    def _fn(self, name):
        return getattr(self._lib, self._prefix + name)

    def _check(self, rc, where):
        if rc != 0:
            raise MMError(rc, self._prefix + where)

    def close(self):
        if self._h:
            self._fn("engine_destroy")(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    # ABI -----------------------------------------------------------------------------
    def lobby_size(self, mode):
        m = self.cfg.modes[mode]
        return int(m.teams * m.team_size)

    def reset(self):
        self._check(self._fn("reset")(self._h), "reset")

    def path_stats(self):
        """mm_path_stats_get as a dict, or None for a library without it (the oracle: it has one way to walk)."""
        fn = getattr(self._lib, self._prefix + "path_stats_get", None)
        if fn is None:
            return None
        fn.argtypes = [C.c_void_p, C.POINTER(MMPathStats)]
        fn.restype = C.c_int
        ps = MMPathStats()
        ps.size = C.sizeof(MMPathStats)
        self._check(fn(self._h, C.byref(ps)), "path_stats_get")
        return ps.as_dict()

    def find_rating_group(self, rating):
        g = C.c_uint32()
        self._check(self._fn("find_rating_group")(C.byref(self.cfg), float(rating), C.byref(g)),
                    "find_rating_group")
        return int(g.value)

    def enqueue(self, rating, cons, group=None):
        rating = np.ascontiguousarray(rating, dtype=np.int32)
        cons = np.ascontiguousarray(cons, dtype=np.uint32)
        assert rating.shape == cons.shape and rating.ndim == 1
        if group is not None:
            group = np.ascontiguousarray(group, dtype=np.uint8)
            assert group.shape == rating.shape
        n = rating.shape[0]
        slots = np.empty(n, dtype=np.uint32)
        st = MMEnqueueStats()
        self._check(self._fn("enqueue")(self._h, n, _ptr(rating), _ptr(cons), _ptr(group),
                                        _ptr(slots), C.byref(st)), "enqueue")
        self.last_enqueue_stats = st.as_dict()
        return slots

    def cancel(self, slots):
        slots = np.ascontiguousarray(slots, dtype=np.uint32)
        self._check(self._fn("cancel")(self._h, slots.shape[0], _ptr(slots)), "cancel")

    def tick(self, mode=0, reuse=False) -> Matches:
        """One tick.  reuse=True copies the match list into buffers the engine object keeps (grown as needed) and
        returns VIEWS of them — valid until the next tick(reuse=True) on this object, like the ABI's own list
        (mm_matches: "readable until the next mm_tick"); a caller that ticks every few milliseconds does not
        want four fresh page-faulting arrays per tick."""
        n = C.c_uint32()
        st = MMStats()
        self._check(self._fn("tick")(self._h, mode, C.byref(n), C.byref(st)), "tick")
        n = int(n.value)
        L = self.lobby_size(mode)
        if reuse:
            have = getattr(self, "_reuse", None)
            if have is None or have[0].size < n * L or have[1].size < n:
                cap = max(n, 1024)
                have = (np.empty(cap * L, np.uint32), np.empty(cap, np.float32), np.empty(cap, np.uint32),
                        np.empty(cap, np.uint32))
                self._reuse = have
            slots, score, group, pass_ = have[0][:n * L].reshape(n, L), have[1][:n], have[2][:n], have[3][:n]
        else:
            slots = np.empty((n, L), dtype=np.uint32)
            score = np.empty(n, dtype=np.float32)
            group = np.empty(n, dtype=np.uint32)
            pass_ = np.empty(n, dtype=np.uint32)
        self._check(self._fn("matches")(self._h, 0, n, _ptr(slots), _ptr(score), _ptr(group),
                                        _ptr(pass_)), "matches")
        return Matches(slots, score, group, pass_, st.as_dict())

    def queue_depth(self, mode=0):
        out = np.zeros(self.cfg.n_groups, dtype=np.uint32)
        self._check(self._fn("queue_depth")(self._h, mode, _ptr(out)), "queue_depth")
        return out

    def queue_slots(self, mode, group):
        """The queue of (mode, group), head first (requeue order, worker.ex:239-248)."""
        n = C.c_uint32(0)
        self._check(self._fn("queue_slots")(self._h, mode, group, C.byref(n), None), "queue_slots")
        out = np.zeros(int(n.value), dtype=np.uint32)
        n2 = C.c_uint32(out.size)
        self._check(self._fn("queue_slots")(self._h, mode, group, C.byref(n2), _ptr(out)), "queue_slots")
        return out[:min(int(n2.value), out.size)]

    def snapshot(self) -> bytes:
        """The whole pool (queues in order, stored lobbies, ActiveUser mirror, slot allocator)."""
        n = C.c_uint64()
        self._check(self._fn("snapshot_size")(self._h, C.byref(n)), "snapshot_size")
        buf = np.empty(int(n.value), dtype=np.uint8)
        w = C.c_uint64()
        self._check(self._fn("snapshot")(self._h, _ptr(buf), C.c_uint64(buf.size), C.byref(w)), "snapshot")
        return buf[:int(w.value)].tobytes()

    def restore(self, blob: bytes):
        buf = np.frombuffer(blob, dtype=np.uint8)
        self._check(self._fn("restore")(self._h, _ptr(buf), C.c_uint64(buf.size)), "restore")

    def lobby_state(self, mode, group):
        n = C.c_uint32()
        slots = np.zeros(MM_MAX_LOBBY, dtype=np.uint32)
        teams = np.zeros(MM_MAX_LOBBY, dtype=np.uint8)
        self._check(self._fn("lobby_state")(self._h, mode, group, C.byref(n), _ptr(slots),
                                            _ptr(teams)), "lobby_state")
        k = int(n.value)
        return slots[:k].copy(), teams[:k].copy()
