#!/usr/bin/env python3
"""Does the pyramid build of time-step k+1 hide behind the walk of time-step k?  Times bench-shaped steps (512 sequences)
sequentially on one stream and pipelined over two streams with double-buffered pyramids.  Diagnostic tool."""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    import torch
    import bench
    from slam_amd import api, capi
    dev = torch.device("cuda", 0)
    nseq, nts, steps = 512, 4, 200
    rng = np.random.default_rng(1)
    frames, jobs, blob, truth = bench.build_time_steps(nseq, nts, 77, rng)
    njobs = nseq * bench.NFEAT
    d_frames = torch.from_numpy(frames).to(dev)
    d_jobs = [torch.from_numpy(j.view(np.uint8).reshape(-1)).to(dev) for j in jobs]
    d_blob = torch.from_numpy(blob).to(dev)
    d_out = [torch.zeros(njobs * bench.TRACK_RES_DT.itemsize, dtype=torch.uint8, device=dev) for _ in range(nts)]
    sA, sB = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(sA)
    ctxA = api.Context(0, stream=sA.cuda_stream)
    ctxB = api.Context(0, stream=sB.cuda_stream)
    W, H = bench.W, bench.H
    fbytes = nseq * W * H
    L = capi.lib()
    mk = lambda ctx: api.ImagePyramid(ctx, None, bench.SUBS, device_ptr=d_frames.data_ptr(), pitch=W, frame_stride=W * H, shape=(nseq, H, W))
    pyrA = mk(ctxA)
    pyrB = [mk(ctxB), mk(ctxB)]

    def walk(ctx, pyr, t):
        capi.check(ctx.handle, L.slam_hip_track_batch_d(ctx.handle, pyr.handle, C.c_void_p(d_jobs[t].data_ptr()), njobs,
                                                        C.c_void_p(d_blob.data_ptr()), blob.size, C.c_void_p(d_out[t].data_ptr())))

    def sequential():
        for k in range(steps):
            t = k % nts
            pyrA.rebuild_d(d_frames.data_ptr() + t * fbytes, W, W * H)
            walk(ctxA, pyrA, t)

    built = [torch.cuda.Event(), torch.cuda.Event()]
    walked = [torch.cuda.Event(), torch.cuda.Event()]

    def pipelined():
        pyrB[0].rebuild_d(d_frames.data_ptr(), W, W * H)
        built[0].record(sB)
        for p in (0, 1):
            walked[p].record(sA)
        for k in range(steps):
            p, t = k & 1, k % nts
            sA.wait_event(built[p])
            walk(ctxA, pyrB[p], t)
            walked[p].record(sA)
            q, tn = (k + 1) & 1, (k + 1) % nts
            sB.wait_event(walked[q])              # the walk that last read this pyramid
            pyrB[q].rebuild_d(d_frames.data_ptr() + tn * fbytes, W, W * H)
            built[q].record(sB)

    for name, fn in (("sequential", sequential), ("pipelined", pipelined), ("sequential", sequential), ("pipelined", pipelined)):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        print("%s: %.4f ms per step" % (name, (time.perf_counter() - t0) / steps * 1e3))
    ref = [d.clone() for d in d_out]
    sequential(); torch.cuda.synchronize()
    print("results equal:", all(torch.equal(a, b) for a, b in zip(ref, d_out)))


if __name__ == "__main__":
    main()
