#!/usr/bin/env python3
"""Times slam_hip_track_batch_d alone on synthetic batches of different shapes (pyramid levels,
template size) to attribute the kernel's time to its phases.  Diagnostic tool, not a benchmark."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def run(subs, tsize, nframes=512, steps=10, thr=0.98, stride=3, mask_cut=True):
    import torch
    from slam_amd import api, capi
    bench.SUBS = subs
    bench.TSIZE = tsize
    bench.THRESHOLD = thr
    bench.STRIDE = stride
    rng = np.random.default_rng(5)
    frames, jobs, blob, truth = bench.build_workload(nframes, 77, rng)
    dev = torch.device("cuda", 0)
    d_frames = torch.from_numpy(frames).to(dev)
    d_jobs = torch.from_numpy(jobs.view(np.uint8).reshape(-1)).to(dev)
    d_blob = torch.from_numpy(blob).to(dev)
    njobs = len(jobs)
    d_out = torch.zeros(njobs * bench.TRACK_RES_DT.itemsize, dtype=torch.uint8, device=dev)
    stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(stream)
    ctx = api.Context(0, stream=stream.cuda_stream)
    pyr = api.ImagePyramid(ctx, None, subs, device_ptr=d_frames.data_ptr(), pitch=bench.W, frame_stride=bench.W * bench.H,
                           shape=(nframes, bench.H, bench.W))
    L = capi.lib()
    if os.environ.get("SLAM_PROBE_LDS"):               # LDS bytes per job (occupancy against chunking)
        capi.check(ctx.handle, L.slam_hip_set_match_lds(ctx.handle, int(os.environ["SLAM_PROBE_LDS"])))

    def go():
        capi.check(ctx.handle, L.slam_hip_track_batch_d(ctx.handle, pyr.handle, C.c_void_p(d_jobs.data_ptr()), njobs,
                                                        C.c_void_p(d_blob.data_ptr()), blob.size, C.c_void_p(d_out.data_ptr())))
    for _ in range(2):
        go()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    for a, b in evs:
        a.record(stream); go(); b.record(stream)
    torch.cuda.synchronize()
    ms = float(np.median([a.elapsed_time(b) for a, b in evs]))
    res = d_out.cpu().numpy().view(bench.TRACK_RES_DT)
    print("subs=%s tsize=%d thr=%.2f stride=%d: %.3f ms / %d jobs = %.1f ns/job, mean levels_run %.2f"
          % (subs, tsize, thr, stride, ms, njobs, ms * 1e6 / njobs, res["levels_run"].mean()), flush=True)
    pyr.close(); ctx.close()


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "size":     # one template size, never stops early
        run([1.0, 2.0, 4.0], int(sys.argv[2]), nframes=256 if int(sys.argv[2]) > 40 else 1024, steps=4, thr=-10.0)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "coarsest":  # every walk stops after the coarsest level
        run([1.0, 2.0, 4.0], 21, nframes=1024, steps=2, thr=10.0)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "one":      # one configuration that never stops early (for counter passes)
        run([1.0, 2.0, 4.0], 21, nframes=1024, steps=2, thr=-10.0)
        sys.exit(0)
    run([1.0, 2.0, 4.0], 21)
    run([1.0, 2.0, 4.0], 21, thr=0.0)      # never stops early
    run([1.0], 21)
    run([1.0, 2.0], 21, thr=0.0)
    run([1.0, 4.0], 21, thr=0.0)
    run([1.0], 11)
    run([1.0], 41)
    run([1.0], 81)
