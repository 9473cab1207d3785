"""The product engine: libmm_engine.so (hand-written HIP for gfx950) behind the C ABI.

There is no CPU fallback.  If the library has not been built, or no GPU is usable, this
module raises — it never routes to the oracle or to any emulation."""
from __future__ import annotations

import ctypes as C
import os

from ._abi import EngineBase, MMConfig, MMEnqueueStats, MMError, bind

_HERE = os.path.dirname(os.path.abspath(__file__))
# MM_ENGINE_LIB: another build of the SAME library (tile geometry experiments, tools/ab_bench.py); never a CPU path
LIB_PATH = os.environ.get("MM_ENGINE_LIB") or os.path.join(_HERE, "csrc", "libmm_engine.so")
_lib = None


class NativeLibraryMissing(ImportError):
    pass


def load_library():
    """dlopen csrc/libmm_engine.so and type its exports.  Loading needs no GPU."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise NativeLibraryMissing(
                "%s not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "or `make -C microservice_matchmaking_amd/csrc` (hipcc --offload-arch=gfx950). "
                "There is no CPU fallback." % LIB_PATH)
        try:
            # torch ships its own libamdhip64.so.7; importing it first makes this process use
            # ONE HIP runtime (ours resolves to the already-loaded SONAME), so torch tensors'
            # device pointers and our stream live in the same runtime.
            import torch  # noqa: F401
        except ImportError:
            pass
        lib = C.CDLL(LIB_PATH)
        bind(lib, "mm_")
        lib.mm_abi_version.restype = C.c_uint32
        lib.mm_strerror.restype = C.c_char_p
        lib.mm_strerror.argtypes = [C.c_int]
        lib.mm_config_default.argtypes = [C.POINTER(MMConfig)]
        lib.mm_config_default.restype = C.c_int
        lib.mm_enqueue_device.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p,
                                          C.POINTER(C.c_uint32), C.POINTER(MMEnqueueStats)]
        lib.mm_enqueue_device.restype = C.c_int
        lib.mm_last_hip_error.argtypes = [C.c_void_p]
        lib.mm_last_hip_error.restype = C.c_int
        _lib = lib
    return _lib


class Engine(EngineBase):
    """One engine = one HIP stream + its device-resident queues and open lobbies."""

    _prefix = "mm_"

    def __init__(self, cfg: MMConfig, tuning=None):
        if Engine._lib is None:
            Engine._lib = load_library()
        super().__init__(cfg, tuning)

    def enqueue_device(self, d_rating, d_cons):
        """rating/cons are CUDA(HIP) torch tensors (int32 / int32-viewed-uint32) already in
        HBM.  Returns the first slot; player i got (first + i) % capacity."""
        n = int(d_rating.numel())
        assert d_rating.is_cuda and d_cons.is_cuda and d_cons.numel() == n
        assert d_rating.element_size() == 4 and d_cons.element_size() == 4
        first = C.c_uint32()
        st = MMEnqueueStats()
        rc = self._lib.mm_enqueue_device(self._h, n, C.c_void_p(d_rating.data_ptr()),
                                         C.c_void_p(d_cons.data_ptr()), C.byref(first), C.byref(st))
        if rc != 0:
            raise MMError(rc, "mm_enqueue_device")
        self.last_enqueue_stats = st.as_dict()
        return int(first.value)

    def last_hip_error(self):
        return int(self._lib.mm_last_hip_error(self._h))
