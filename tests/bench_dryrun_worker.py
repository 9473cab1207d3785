"""One rank of a DRY RUN of `bench.py --gpus N` without GPUs (tests/test_bench_line.py): the
engine is the fiber-shim build of the kernel source, torch.cuda is stubbed and RCCL is replaced
by gloo, so that the rank / barrier / all-reduce / one-JSON-line logic of the multi-GPU launch
is executed once before the driver runs it on real hardware.  Prints nothing of its own."""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

import microservice_matchmaking_amd as pkg  # noqa: E402
from test_bench_line import DryEngine  # noqa: E402

torch.cuda.is_available = lambda: True
torch.cuda.set_device = lambda d: None
torch.cuda.device_count = lambda: 8
torch.cuda.synchronize = lambda *a: None
torch.Tensor.cuda = lambda self, *a, **k: self
_tensor = torch.tensor
torch.tensor = lambda *a, **k: _tensor(*a, **{x: y for x, y in k.items() if x != "device"})
_init = dist.init_process_group
dist.init_process_group = lambda backend=None, **k: _init("gloo")
pkg.Engine = DryEngine

import bench  # noqa: E402

# `bench.py --gpus N` without a launcher re-executes itself once per rank: in the dry run, this worker
bench.SELF_CMD = [sys.executable, os.path.abspath(__file__)]

if __name__ == "__main__":
    sys.argv = ["bench.py"] + sys.argv[1:]
    if "LOCAL_RANK" in os.environ:
        os.environ["LOCAL_RANK"] = "0"               # the shim has one device
    bench.main()
