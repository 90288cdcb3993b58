"""The RCCL leg of the N > 1 path, executed: torch.distributed backend "nccl" (= RCCL on ROCm) with a group of one rank
on cuda:0 and device tensors through the same helpers bench.py uses for its reductions and the validation gather.  (Two
ranks on one device are refused by RCCL; more devices need a multi-GPU node.)  Runs in a child process so that the
process group never outlives the test."""
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu

CHILD = r"""
import os, sys
import numpy as np
sys.path.insert(0, %r)
import torch
import torch.distributed as dist
from slam_amd import parallel
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%%d" %% int(sys.argv[1]), rank=0, world_size=1, device_id=dev)
assert dist.get_backend() == "nccl"
t, c = parallel.reduce_max_sum(1.25, 7.0, dev)
assert (t, c) == (1.25, 7.0), (t, c)
v = parallel.reduce_max_vector([0.5, 2.0, 1.0], dev)
assert v == [0.5, 2.0, 1.0], v
rec = np.zeros(5, np.dtype([("x", "<i4"), ("r", "<f4")]))
rec["x"] = np.arange(5); rec["r"] = 0.5 * np.arange(5)
out = parallel.gather_records(rec, dev)
assert out.dtype == rec.dtype and np.array_equal(out, rec)
empty = parallel.gather_records(rec[:0], dev)
assert len(empty) == 0
dist.barrier()
torch.cuda.synchronize()
dist.destroy_process_group()
print("rccl ok")
""" % ROOT


def test_rccl_collectives_run_with_device_tensors():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    out = subprocess.run([sys.executable, "-c", CHILD, str(port)], capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    assert "rccl ok" in out.stdout
