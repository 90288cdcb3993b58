#!/usr/bin/env python3
"""Phase breakdown of the track kernel from in-kernel s_memtime stamps (diagnostic build only:
`make -C slam_amd/csrc stamps`, run with SLAM_HIP_LIB=slam_amd/lib/libslam_hip_stamps.so).
Prints the median shader cycles between consecutive stamps, per stamp-id transition."""
import collections
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
    nframes = 512
    if len(sys.argv) > 1:                               # template size (default: the bench's)
        bench.TSIZE = int(sys.argv[1])
        bench.THRESHOLD = -10.0                         # never stops early
        nframes = 128 if bench.TSIZE > 40 else 512
    rng = np.random.default_rng(5)
    frames, jobs, blob, truth = bench.build_time_steps(nframes, 1, 77, rng)
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
    trans = collections.defaultdict(list)
    totals = []
    for j in range(0, njobs, 7):
        row = st[j]
        row = row[row != 0]
        ids = (row & np.uint64(0xFF)).astype(int)
        ts = (row >> np.uint64(8)).astype(np.int64)
        totals.append(ts[-1] - ts[0])
        lvl = 0
        for a in range(len(ids) - 1):
            if ids[a] == 2:
                lvl += 1
            trans[(lvl, ids[a], ids[a + 1])].append(ts[a + 1] - ts[a])
    print("jobs sampled %d, median total cycles per job %d (s_memtime ticks at 100 MHz? see ratio below)" % (len(totals), np.median(totals)))
    tot = 0
    for k in sorted(trans):
        v = np.array(trans[k])
        print("level#%d  %2d -> %2d : n/job %.2f  median %8.0f  mean %8.0f" % (k[0], k[1], k[2], len(v) / len(totals), np.median(v), v.mean()))
        tot += v.sum() / len(totals)
    print("sum of means per job: %.0f" % tot)
    pyr.close(); ctx.close()


if __name__ == "__main__":
    main()
