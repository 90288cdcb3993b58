"""N>1 path on CPU: two gloo ranks shard a frame batch, each tracks its own frames (with the oracle
standing in for the GPU), and the gathered results equal the single-process results."""
import os
import socket
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

RES_DT = np.dtype([("frame", "<i4"), ("x", "<i4"), ("y", "<i4"), ("response", "<f4"), ("levels_run", "<i4")])


def _workload():
    from slam_amd import synth
    rng = np.random.default_rng(3)
    w, h, subs = 160, 120, [1.0, 2.0]
    frames, jobs = [], []
    for f in range(5):                      # 5 frames over 2 ranks: uneven shards
        base = synth.frame(w, h, 50 + f)
        frames.append(base)
        for (cx, cy) in synth.feature_grid(w, h, 6, 25, rng):
            proj = synth.cut_template(base, cx, cy, 9, 9)
            jobs.append(dict(frame=f, proj=proj, proj_mask=np.ones_like(proj), px=cx + 1.3, py=cy - 0.6,
                             threshold=0.9, stride=2))
    return frames, subs, jobs


def _track_frames(frames, subs, jobs, lo, hi):
    from oracle import pyoracle as O
    out = []
    for f in range(lo, hi):
        levels = O.pyramid(frames[f], subs)
        for j in jobs:
            if j["frame"] == f:
                r = O.track(levels, subs, j["proj"], j["proj_mask"], j["px"], j["py"], j["threshold"], j["stride"])
                out.append((f, r["x"], r["y"], r["response"], r["levels_run"]))
    return np.array(out, RES_DT)


def _rank_main(rank, world, port, q):
    import torch.distributed as dist
    from slam_amd import parallel
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    frames, subs, jobs = _workload()
    lo, hi = parallel.shard_range(len(frames), rank, world)
    local = _track_frames(frames, subs, jobs, lo, hi)
    allres = parallel.gather_records(local)
    tmax, csum = parallel.reduce_max_sum(1.0 + rank, len(local))
    dist.barrier()
    if rank == 0:
        q.put((allres.tobytes(), tmax, csum, (lo, hi)))
    dist.destroy_process_group()


def test_two_rank_gloo_sharding_matches_single_process():
    import torch.multiprocessing as mp
    from oracle import pyoracle as O
    from slam_amd import parallel
    O.build()
    assert [parallel.shard_range(5, r, 2) for r in range(2)] == [(0, 2), (2, 5)]
    assert [parallel.shard_range(4096, r, 8) for r in range(8)] == [(512 * r, 512 * (r + 1)) for r in range(8)]
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_rank_main, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    raw, tmax, csum, r0 = q.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    frames, subs, jobs = _workload()
    single = _track_frames(frames, subs, jobs, 0, len(frames))
    got = np.frombuffer(raw, RES_DT)
    assert np.array_equal(got, single)          # rank order == frame order: no re-sorting needed
    assert tmax == 2.0 and csum == len(single) and r0 == (0, 2)
