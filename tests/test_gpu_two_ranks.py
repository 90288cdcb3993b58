"""The N > 1 path on the HIP library: two rank processes on one GPU (gloo for the small reductions), each tracking its
own shard of the sequences through the C ABI; the gathered records equal the single-process HIP results, and
`bench.py --gpus 2` starts its own ranks and prints one line.  RCCL itself needs a multi-GPU node and is not run here."""
import ctypes as C
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

pytestmark = pytest.mark.gpu

NSEQ, NTS = 6, 1


def _track_shard(lo, hi):
    """HIP results (records) of sequences [lo, hi) of the bench workload's first time-step."""
    import bench
    from slam_amd import api, capi
    rng = np.random.default_rng(11)
    frames, jobs, blob, _ = bench.build_time_steps(NSEQ, NTS, 4242, rng)
    frames, jobs = frames[0, lo:hi], jobs[0][lo * bench.NFEAT:hi * bench.NFEAT].copy()
    jobs["frame"] -= lo
    with api.Context(0) as ctx:
        pyr = api.ImagePyramid(ctx, frames, bench.SUBS)
        res = np.zeros(len(jobs), bench.TRACK_RES_DT)
        capi.check(ctx.handle, capi.lib().slam_hip_track_batch(ctx.handle, pyr.handle, jobs.ctypes.data, len(jobs),
                                                               blob.ctypes.data, blob.size, res.ctypes.data))
        pyr.close()
    return res


def _rank_main(rank, world, port, q):
    import torch.distributed as dist
    from slam_amd import parallel
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = parallel.shard_range(NSEQ, rank, world)
    local = _track_shard(lo, hi)
    allres = parallel.gather_records(local)
    tmax, csum = parallel.reduce_max_sum(1.0 + rank, len(local))
    dist.barrier()
    if rank == 0:
        q.put((allres.tobytes(), tmax, csum))
    dist.destroy_process_group()


def test_two_hip_ranks_gather_the_single_process_result():
    import torch.multiprocessing as mp
    import bench
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_rank_main, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    raw, tmax, csum = q.get(timeout=300)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    single = _track_shard(0, NSEQ)
    got = np.frombuffer(raw, bench.TRACK_RES_DT)
    assert (single["status"] == 0).all() and (single["levels_run"] >= 1).all()
    assert np.array_equal(got, single)
    assert tmax == 2.0 and csum == len(single)


def test_bench_starts_its_own_ranks():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--rehearse-one-device",
                          "--sequences-per-gpu", "16", "--time-steps", "2", "--steps", "2", "--warmup", "1", "--cpu-procs", "0"],
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1                                  # rank 0 alone prints
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["config"]["sequences"] == 32 and line["value"] > 0
    assert line["config"]["parity"] and line["cpu_baseline"] is None
