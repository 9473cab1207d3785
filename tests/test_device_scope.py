"""HIP's current device is per-thread state and the callers the ABI is shaped for do not keep a
thread (dirty NIFs run on whichever dirty scheduler is free).  Every entry point that touches the
device must select the engine's device itself and put the caller's back.  Checked on the
fiber-shim build, whose hipSetDevice/hipGetDevice keep a thread-local ordinal and count calls."""
import threading


from emu_engine import EmuEngine, load
from microservice_matchmaking_amd.config import make_config, mode_1v1
from microservice_matchmaking_amd.synth import make_pool


def test_every_entry_point_selects_the_engines_device_and_restores_the_callers():
    lib = load()
    cfg = make_config([mode_1v1(window=25, region_filter=True)], capacity=4096)
    rating, cons = make_pool(2000, seed=3)
    lib.emu_set_current_device(3)                  # the host program works on another GPU
    try:
        eng = EmuEngine(cfg)                       # cfg.device = 0
        assert lib.emu_current_device() == 3       # create put it back

        def scoped(fn, *a):
            before = lib.emu_set_device_calls()
            out = fn(*a)
            assert lib.emu_current_device() == 3, fn.__name__
            assert lib.emu_set_device_calls() >= before + 2, fn.__name__   # selected 0, restored 3
            return out

        slots = scoped(eng.enqueue, rating, cons)
        scoped(eng.cancel, slots[:10])
        m = scoped(eng.tick, 0)
        assert len(m) > 0
        scoped(eng.queue_depth, 0)
        scoped(eng.lobby_state, 0, 0)
        blob = scoped(eng.snapshot)
        scoped(eng.restore, blob)
        scoped(eng.reset)
        scoped(eng.close)
    finally:
        lib.emu_set_current_device(0)


def test_calls_from_another_thread_need_no_setup_and_no_switch_on_the_same_device():
    lib = load()
    cfg = make_config([mode_1v1(window=25, region_filter=True)], capacity=4096)
    rating, cons = make_pool(2000, seed=4)
    eng = EmuEngine(cfg)
    result = {}

    def worker():                                  # a fresh thread: its current device is 0 = the engine's
        before = lib.emu_set_device_calls()
        eng.enqueue(rating, cons)
        result["n"] = len(eng.tick(0))
        result["switches"] = lib.emu_set_device_calls() - before

    t = threading.Thread(target=worker)
    t.start()
    t.join()
    eng.close()
    assert result["n"] > 0 and result["switches"] == 0
