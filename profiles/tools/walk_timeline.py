#!/usr/bin/env python3
"""Occupancy over time of one launch of the lean walk kernel on the bench's workload, from the in-kernel s_memtime
stamps of the diagnostic build (`make -C slam_amd/csrc stamps FLAGS_EXTRA=-DSLAM_STAMPS_REALTIME`; run with SLAM_HIP_LIB=slam_amd/lib/libslam_hip_stamps.so):
how many walks are in flight in every twentieth of the launch, and how long a walk lives depending on when it starts."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

SLOTS = 64


def main():
    import torch
    from slam_amd import api, capi
    nframes = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    frames, jobs, blob, _ = bench.build_time_steps(nframes, 1, 77, np.random.default_rng(5))
    frames, jobs = frames[0], jobs[0]
    dev = torch.device("cuda", 0)
    d_frames = torch.from_numpy(frames).to(dev)
    d_jobs = torch.from_numpy(jobs.view(np.uint8).reshape(-1)).to(dev)
    d_blob = torch.from_numpy(blob).to(dev)
    njobs = len(jobs)
    d_out = torch.zeros(njobs * bench.TRACK_RES_DT.itemsize, dtype=torch.uint8, device=dev)
    d_st = torch.zeros(njobs * SLOTS, dtype=torch.int64, device=dev)
    stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(stream)
    ctx = api.Context(0, stream=stream.cuda_stream)
    pyr = api.ImagePyramid(ctx, None, bench.SUBS, device_ptr=d_frames.data_ptr(), pitch=bench.W,
                           frame_stride=bench.W * bench.H, shape=(nframes, bench.H, bench.W))
    L = capi.lib()
    raw = C.CDLL(capi.LIB_PATH)
    raw.slam_hip_debug_set_stamps.argtypes = [C.c_void_p]

    def go():
        capi.check(ctx.handle, L.slam_hip_track_batch_d(ctx.handle, pyr.handle, C.c_void_p(d_jobs.data_ptr()), njobs,
                                                        C.c_void_p(d_blob.data_ptr()), blob.size, C.c_void_p(d_out.data_ptr())))
    go(); go()
    torch.cuda.synchronize()
    assert raw.slam_hip_debug_set_stamps(C.c_void_p(d_st.data_ptr())) == 0
    go()
    torch.cuda.synchronize()
    st = d_st.cpu().numpy().view(np.uint64).reshape(njobs, SLOTS)
    ts = (st >> np.uint64(8)).astype(np.int64)
    ts[st == 0] = -1
    first = ts[:, 0].copy()
    last = ts.max(axis=1)
    b = first.min()
    first -= b; last -= b
    span = last.max()
    life = last - first
    print("walks %d, launch span %.1f us (100 MHz counter)" % (njobs, span / 100.0))
    print("walk lifetime in us: median %.1f, 10%% %.1f, 90%% %.1f" % (np.median(life) / 100.0, np.percentile(life, 10) / 100.0, np.percentile(life, 90) / 100.0))
    edges = np.linspace(0, span, 26)
    for a, b in zip(edges[:-1], edges[1:]):
        mid = 0.5 * (a + b)
        act = int(np.sum((first <= mid) & (last >= mid)))
        started = (first >= a) & (first < b)
        lt = np.median(life[started]) if started.any() else float("nan")
        print("t = %5.1f%% : %5d walks in flight, %5d start here (median lifetime %5.1f us), %5d end here"
              % (100.0 * mid / span, act, int(started.sum()), lt / 100.0, int(np.sum((last >= a) & (last < b)))))
    pyr.close(); ctx.close()


if __name__ == "__main__":
    main()
