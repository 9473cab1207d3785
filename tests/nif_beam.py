"""A pretend BEAM for tests: loads native/mm_nif.c (compiled unmodified against the stand-in
erl_nif of tests/nif/) and calls its NIF table with Elixir-shaped terms.  Test infrastructure.

Python value <-> term:  bytes <-> binary, int <-> integer, float <-> float, str <-> atom,
list <-> list, tuple <- tuple, Resource <-> resource handle; a charlist comes back as Charlist,
a raised badarg as the BadArg exception (the BEAM would raise ArgumentError in the caller)."""
import ctypes as C
import os
import subprocess

_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "nif")
T_ATOM, T_INT, T_FLOAT, T_BIN, T_TUPLE, T_CONS, T_NIL, T_RES, T_STR, T_EXC = range(1, 11)
DIRTY_CPU, DIRTY_IO = 1, 2


class BadArg(Exception):
    pass


class Charlist(str):
    pass


class Resource:
    def __init__(self, term):
        self.term = term


class Beam:
    """One loaded NIF library + one process heap (env) that lives until gc()."""

    def __init__(self, backend):
        from emu_engine import load as build_emu
        build_emu()                                               # libmm_engine_emu.so must exist first
        subprocess.check_call(["make", "-C", _DIR, "--no-print-directory"], stdout=subprocess.DEVNULL)
        if backend == "hip":
            # same loading order as the product binding (torch first: one HIP runtime per process)
            from microservice_matchmaking_amd.engine import load_library
            load_library()
        L = self.L = C.CDLL(os.path.join(_DIR, "mm_nif_%s.so" % backend))
        vp, term = C.c_void_p, C.c_size_t
        for name, res, args in (
                ("nif_init", vp, []), ("mock_env_new", vp, []), ("mock_env_free", None, [vp]),
                ("mock_live_resources", C.c_long, []), ("mock_load", C.c_int, [vp, vp]),
                ("mock_nif_count", C.c_int, [vp]), ("mock_module", C.c_char_p, [vp]),
                ("mock_nif_name", C.c_char_p, [vp, C.c_int]), ("mock_nif_arity", C.c_uint, [vp, C.c_int]),
                ("mock_nif_flags", C.c_uint, [vp, C.c_int]),
                ("mock_call", term, [vp, C.c_int, vp, C.c_int, C.POINTER(term)]),
                ("mock_mk_binary", term, [vp, C.c_char_p, C.c_size_t]), ("mock_mk_int", term, [vp, C.c_longlong]),
                ("mock_mk_float", term, [vp, C.c_double]), ("mock_mk_atom", term, [C.c_char_p]),
                ("mock_mk_list", term, [vp, C.c_uint, C.POINTER(term)]), ("mock_tag", C.c_int, [term]),
                ("mock_tuple_arity", C.c_uint, [term]), ("mock_tuple_elem", term, [term, C.c_uint]),
                ("mock_bin_data", vp, [term]), ("mock_bin_size", C.c_size_t, [term]),
                ("mock_int_value", C.c_longlong, [term]), ("mock_uint_value", C.c_ulonglong, [term]),
                ("mock_atom_name", C.c_char_p, [term]), ("mock_str", C.c_char_p, [term])):
            f = getattr(L, name)
            f.restype, f.argtypes = res, args
        self.entry = L.nif_init()
        self.env = L.mock_env_new()
        assert L.mock_load(self.entry, self.env) == 0, "load/3 of the NIF failed"
        self.table = {}
        for i in range(L.mock_nif_count(self.entry)):
            self.table[(L.mock_nif_name(self.entry, i).decode(), L.mock_nif_arity(self.entry, i))] = (
                i, L.mock_nif_flags(self.entry, i))
        self.module = L.mock_module(self.entry).decode()

    # terms ---------------------------------------------------------------------------
    def to_term(self, v):
        L = self.L
        if isinstance(v, Resource):
            return v.term
        if isinstance(v, (bytes, bytearray, memoryview)):
            b = bytes(v)
            return L.mock_mk_binary(self.env, b, len(b))
        if isinstance(v, bool):
            return L.mock_mk_atom(b"true" if v else b"false")
        if isinstance(v, int):
            return L.mock_mk_int(self.env, v)
        if isinstance(v, float):
            return L.mock_mk_float(self.env, v)
        if v is None:
            return L.mock_mk_atom(b"nil")
        if isinstance(v, str):
            return L.mock_mk_atom(v.encode())
        if isinstance(v, list):
            arr = (C.c_size_t * max(1, len(v)))(*[self.to_term(x) for x in v])
            return L.mock_mk_list(self.env, len(v), arr)
        raise TypeError(type(v))

    def from_term(self, t):
        L = self.L
        tag = L.mock_tag(t)
        if tag == T_ATOM:
            return L.mock_atom_name(t).decode()
        if tag == T_INT:
            return int(L.mock_int_value(t)) if L.mock_int_value(t) >= 0 or L.mock_uint_value(t) >> 63 == 0 \
                else int(L.mock_int_value(t))
        if tag == T_BIN:
            return C.string_at(L.mock_bin_data(t), L.mock_bin_size(t))
        if tag == T_TUPLE:
            return tuple(self.from_term(L.mock_tuple_elem(t, i)) for i in range(L.mock_tuple_arity(t)))
        if tag == T_RES:
            return Resource(t)
        if tag == T_STR:
            return Charlist(L.mock_str(t).decode("latin-1"))
        if tag == T_EXC:
            raise BadArg()
        raise TypeError("term tag %d" % tag)

    # Matchmaking.Search.Engine.<name>(args...) -----------------------------------------
    def call(self, name, *args):
        i, _ = self.table[(name, len(args))]
        argv = (C.c_size_t * max(1, len(args)))(*[self.to_term(a) for a in args])
        return self.from_term(self.L.mock_call(self.entry, i, self.env, len(args), argv))

    def gc(self):
        """Drop every term of the process (resources without other owners run their destructor)."""
        self.L.mock_env_free(self.env)
        self.env = self.L.mock_env_new()

    def live_resources(self):
        return int(self.L.mock_live_resources())
