"""The C-ABI library loads and exports every symbol include/mm_engine.h declares
(no compute calls: there is no GPU in the `-m "not gpu"` environment)."""
import ctypes as C
import os
import re
import subprocess

import pytest

from conftest import ROOT
from microservice_matchmaking_amd import MMError, make_config, mode_1v1
from microservice_matchmaking_amd._abi import MMConfig, MMEnqueueStats, MMModeConfig, MMPathStats, MMStats
from microservice_matchmaking_amd.engine import LIB_PATH, load_library

HEADERS = [os.path.join(ROOT, "include", h) for h in ("mm_engine.h", "mm_codec.h")]


def declared_functions():
    names = set()
    for h in HEADERS:
        src = re.sub(r"/\*.*?\*/", "", open(h).read(), flags=re.S)
        names |= set(re.findall(r"\b(mm_[a-z_0-9]+)\s*\(", src))
    return sorted(names)


@pytest.fixture(scope="module")
def lib():
    subprocess.check_call(["make", "-C", os.path.dirname(LIB_PATH), "--no-print-directory"],
                          stdout=subprocess.DEVNULL)
    return load_library()


def test_every_declared_symbol_is_exported(lib):
    names = declared_functions()
    assert len(names) >= 15, names
    for n in names:
        assert hasattr(lib, n), "libmm_engine.so does not export %s" % n


def test_oracle_exports_the_mirrored_abi(oracle_cls):
    from oracle.oracle import load
    olib = load()
    for n in declared_functions():
        if n in ("mm_abi_version", "mm_strerror", "mm_config_default", "mm_enqueue_device",
                 "mm_last_hip_error", "mm_path_stats_get", "mm_snapshot_size", "mm_snapshot", "mm_restore", "mm_decode_players", "mm_encode_lobby",
                 "mm_engine_create_ex", "mm_tuning_default", "mm_tuning_set", "mm_tuning_name", "mm_tuning_get"):
            continue
        assert hasattr(olib, "mo_" + n[3:]), n


def test_struct_layouts_match_the_header(lib):
    # sizes the C compiler gives the header's structs, via a tiny probe program
    probe = r'''
#include <stdio.h>
#include "mm_engine.h"
int main(void){printf("%zu %zu %zu %zu %zu\n", sizeof(mm_config), sizeof(mm_mode_config), sizeof(mm_stats), sizeof(mm_enqueue_stats), sizeof(mm_path_stats));return 0;}
'''
    exe = "/tmp/mm_abi_probe"
    subprocess.run(["gcc", "-x", "c", "-", "-I", os.path.join(ROOT, "include"), "-o", exe],
                   input=probe.encode(), check=True)
    sizes = [int(x) for x in subprocess.check_output([exe]).split()]
    assert sizes == [C.sizeof(MMConfig), C.sizeof(MMModeConfig), C.sizeof(MMStats), C.sizeof(MMEnqueueStats), C.sizeof(MMPathStats)]


def test_tuning_record_by_name_and_by_layout(lib, monkeypatch):
    """mm_tuning (include/mm_engine.h): every uint32_t field of the header's struct has its name in mm_tuning_name, in the
    struct's order; mm_tuning_default fills as many bytes as the caller has room for; mm_tuning_set checks names and
    ranges; the MM_* environment supplies defaults only, and a value that is not one is reported and ignored (ADVICE r05:
    MM_PAIR_TILE=2048 used to be a silent no-op).  No GPU needed: nothing here creates an engine."""
    import re
    from microservice_matchmaking_amd import Engine
    Engine._lib = lib
    hdr = open(os.path.join(ROOT, "include", "mm_engine.h")).read()
    body = hdr[hdr.index("typedef struct mm_tuning {"):hdr.index("} mm_tuning;")]
    fields = re.findall(r"^\s*uint32_t\s+(\w+);", body, re.M)
    assert fields[0] == "size" and fields[1:] == Engine.tuning_names() and len(fields) >= 30
    probe = '#include <stdio.h>\n#include "mm_engine.h"\nint main(void){printf("%zu\\n", sizeof(mm_tuning));return 0;}\n'
    exe = "/tmp/mm_tuning_probe"
    subprocess.run(["gcc", "-x", "c", "-", "-I", os.path.join(ROOT, "include"), "-o", exe], input=probe.encode(), check=True)
    assert int(subprocess.check_output([exe])) == 4 * len(fields)
    for k in list(os.environ):
        if k.startswith("MM_"):
            monkeypatch.delenv(k)
    d = Engine.tuning_defaults()
    assert (d["pair_persist"], d["pair_ptiles"], d["pair_pbatch"], d["team_f2"], d["team_late"], d["team_fix_max"]) == (1, 32, 48, 32, 6, 0xFFFFFFFF)
    # a short record of an older caller: only what fits is filled, `size` says how much
    rec = (C.c_uint32 * 8)(12, 7, 7, 7, 7, 7, 7, 7)
    lib.mm_tuning_default.argtypes, lib.mm_tuning_default.restype = [C.c_void_p], C.c_int
    assert lib.mm_tuning_default(rec) == 0 and list(rec) == [12, d["force_generic"], d["debug"], 7, 7, 7, 7, 7]
    rec = Engine.tuning_record({"team_late": 0, "pair_nxseg": 512})
    names = Engine.tuning_names()
    assert rec[1 + names.index("team_late")] == 0 and rec[1 + names.index("pair_nxseg")] == 512
    for bad, code in (({"pair_nxseg": 500}, -8), ({"pair_ptiles": 0}, -8), ({"team_cap": 5000}, -8), ({"pair_fused": 0}, -1)):
        with pytest.raises(MMError) as ei:
            Engine.tuning_record(bad)
        assert ei.value.status == code, bad
    # the environment is the default, nothing more — and only for values of the field
    monkeypatch.setenv("MM_TEAM_LATE", "11")
    monkeypatch.setenv("MM_PAIR_TILE", "max")
    monkeypatch.setenv("MM_PAIR_PTILES", "2048")             # outside 1..32: reported on stderr, ignored
    d2 = Engine.tuning_defaults()
    assert (d2["team_late"], d2["pair_tile_fixed"], d2["pair_ptiles"]) == (11, 1, 32)
    assert Engine.tuning_record({"team_late": 3})[1 + names.index("team_late")] == 3


def test_every_tuning_field_round_trips_through_set_by_name(lib):
    """mm_tuning_set writes exactly the named word of the record and nothing else, for every field the library lists, at
    both ends of what it accepts (the defaults are always acceptable: a record of defaults set field by field is itself)."""
    from microservice_matchmaking_amd import Engine
    Engine._lib = lib
    names, base = Engine.tuning_names(), Engine.tuning_record()
    for i, n in enumerate(names):
        rec = Engine.tuning_record({n: base[1 + i]})
        assert list(rec) == list(base), n
        for v in (0, 1, 2, 64, 0xFFFFFFFF):
            try:
                rec = Engine.tuning_record({n: v})
            except MMError as ex:
                assert ex.status == -8, (n, v)          # MM_ERR_RANGE, never anything else for a known name
                continue
            diff = [k for k in range(len(rec)) if rec[k] != base[k]]
            assert diff in ([], [1 + i]) and rec[1 + i] == v, (n, v)


def test_library_level_calls_need_no_gpu(lib):
    assert lib.mm_abi_version() == 1
    assert lib.mm_strerror(0) == b"ok"
    assert b"device" in lib.mm_strerror(-2)
    cfg = MMConfig()
    assert lib.mm_config_default(C.byref(cfg)) == 0
    assert cfg.n_groups == 7 and cfg.default_group == 4 and cfg.groups[6].to == 5000
    g = C.c_uint32()
    for rating, want in ((1499, 0), (1500, 1), (5001, 4), (1499.5, 4), (float("nan"), 4)):
        assert lib.mm_find_rating_group(C.byref(cfg), rating, C.byref(g)) == 0
        assert g.value == want


def test_bad_config_is_rejected_before_touching_a_device(lib):
    cfg = make_config([mode_1v1()])
    cfg.abi_version = 99
    h = C.c_void_p()
    assert lib.mm_engine_create(C.byref(cfg), C.byref(h)) == -7          # MM_ERR_ABI
    cfg = make_config([mode_1v1()])
    cfg.modes[0].role_quota[0] = 2                                        # sums to 2 != team_size
    assert lib.mm_engine_create(C.byref(cfg), C.byref(h)) == -1
    assert lib.mm_engine_create(None, C.byref(h)) == -1
    lib.mm_engine_destroy(None)                                           # NULL-safe


def test_product_fails_loudly_without_a_gpu(lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from microservice_matchmaking_amd import Engine
    with pytest.raises(MMError) as ei:
        Engine(make_config([mode_1v1()]))
    assert ei.value.status == -2   # MM_ERR_NO_DEVICE — never a silent CPU path
