#!/usr/bin/env python3
"""Launches the point-walk kernels a few times on the bench's workload (no oracle, no timing): the target of counter
passes and of the diagnostic builds that end every wave at a phase boundary.  usage: walk_probe.py [sequences] [launches]"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    import torch
    from slam_amd import api, capi
    nseq = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    frames, jobs, blob, _ = bench.build_time_steps(nseq, 1, 77, np.random.default_rng(5))
    frames, jobs = frames[0], jobs[0]
    dev = torch.device("cuda", 0)
    d_frames = torch.from_numpy(frames).to(dev)
    d_jobs = torch.from_numpy(jobs.view(np.uint8).reshape(-1)).to(dev)
    d_blob = torch.from_numpy(blob).to(dev)
    d_out = torch.zeros(len(jobs) * bench.TRACK_RES_DT.itemsize, dtype=torch.uint8, device=dev)
    stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(stream)
    ctx = api.Context(0, stream=stream.cuda_stream)
    pyr = api.ImagePyramid(ctx, None, bench.SUBS, device_ptr=d_frames.data_ptr(), pitch=bench.W,
                           frame_stride=bench.W * bench.H, shape=(nseq, bench.H, bench.W))
    L = capi.lib()
    for _ in range(reps):
        capi.check(ctx.handle, L.slam_hip_track_batch_d(ctx.handle, pyr.handle, C.c_void_p(d_jobs.data_ptr()), len(jobs),
                                                        C.c_void_p(d_blob.data_ptr()), blob.size, C.c_void_p(d_out.data_ptr())))
    torch.cuda.synchronize()
    pyr.close(); ctx.close()


if __name__ == "__main__":
    main()
