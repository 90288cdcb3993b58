"""`python bench.py --gpus N` must start its own ranks: the parent builds a torch.distributed.run command for N
processes on 127.0.0.1, never imports torch (so it cannot touch a GPU) and passes the children's exit code on."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def test_rank_command_is_one_process_per_gpu_on_localhost():
    argv = ["--gpus", "8", "--steps", "5", "--warmup", "2"]
    cmd = bench.rank_launch_command(bench.parse_args(argv), argv, 29511)
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and "--nproc-per-node=8" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29511"
    i = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[i + 1:] == argv                      # the ranks see the caller's own arguments


def test_parent_launches_instead_of_running(monkeypatch):
    seen = {}

    def fake_call(cmd, env=None):
        seen["cmd"], seen["env"] = cmd, env
        return 7
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(bench.subprocess, "call", fake_call)
    with pytest.raises(SystemExit) as e:
        bench.main(["--gpus", "2", "--steps", "1"])
    assert e.value.code == 7                                    # the ranks' exit code is the command's
    assert "--nproc-per-node=2" in seen["cmd"]
    assert seen["env"].get("HSA_ENABLE_IPC_MODE_LEGACY") == "0"


def test_default_is_the_c5_shard():
    a = bench.parse_args([])
    assert (a.gpus, a.sequences_per_gpu) == (1, 512)            # 8 GPUs x 512 = the 4096 sequences of C5
    from slam_amd import parallel
    assert [parallel.shard_range(512 * 8, r, 8) for r in (0, 7)] == [(0, 512), (3584, 4096)]


def test_launcher_process_never_imports_torch():
    """The launching parent must stay clear of the GPU: importing bench and building the launch command loads no torch."""
    code = ("import sys; sys.path.insert(0, %r); import bench; a = ['--gpus', '4']; "
            "bench.rank_launch_command(bench.parse_args(a), a, 1); "
            "assert 'torch' not in sys.modules, 'torch imported by the launcher'" % ROOT)
    subprocess.check_call([sys.executable, "-c", code])
