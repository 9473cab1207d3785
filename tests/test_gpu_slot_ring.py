"""mm_enqueue's slot allocation on the real GPU (k_bucket_scatter with a host-picked slot list).
Added after the round's last GPU run: kept in a file of its own, after test_gpu_parity.py in
collection order, so that `pytest -x` reports the long-standing parity tests first."""
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu_cls():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU; there is no CPU fallback"
    from microservice_matchmaking_amd import Engine
    return Engine


def test_gpu_stream_laps_the_slot_ring_around_waiting_players(gpu_cls, oracle_cls):
    """mm_enqueue hands out the next FREE slots in ring order: a player that waits for hours keeps
    its slot while the ring laps it (k_bucket_scatter with a host-picked slot list)."""
    from helpers import run_wrapping_stream
    laps, stepped = run_wrapping_stream(gpu_cls, oracle_cls, capacity=4096, ticks=150, per_tick=600)
    assert laps > 10 and stepped > 20, (laps, stepped)


def test_gpu_enqueue_device_rejects_leave_their_slots_free(gpu_cls):
    import ctypes as C
    import torch
    from test_emu_kernels import enqueue_device_rejects_leave_their_slots_free

    def to_device(rating, cons):
        d_r, d_c = torch.from_numpy(rating).cuda(), torch.from_numpy(cons.view("int32")).cuda()
        return C.c_void_p(d_r.data_ptr()), C.c_void_p(d_c.data_ptr()), (d_r, d_c)
    enqueue_device_rejects_leave_their_slots_free(gpu_cls, to_device)
