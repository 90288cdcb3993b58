#!/usr/bin/env python3
"""Captures four pipelined bench steps (pyramid of step k + 1 on a second stream beside the walk of step k) in one HIP
graph through torch.cuda.CUDAGraph and replays it: the `_d` entry points only enqueue on their context's stream, so
they can be captured once their queues have their size (a few eager calls first).  Prints the time per step of the
replayed graph and whether its results equal the eager run's.  Diagnostic tool: the graph is not faster than the eager
loop (0.285 against 0.280 ms per step), the step is bound by its kernels, not by their launches."""
import ctypes as C, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
import torch
from slam_amd import api, capi
nseq, nts = 512, 4
dev = torch.device("cuda", 0)
frames, jobs, blob, truth = bench.build_time_steps(nseq, nts, 0x51A40002, np.random.default_rng(0x51A40002))
njobs = nseq * bench.NFEAT
W, H = bench.W, bench.H
d_frames = torch.from_numpy(frames).to(dev)
d_jobs = [torch.from_numpy(j.view(np.uint8).reshape(-1)).to(dev) for j in jobs]
d_blob = torch.from_numpy(blob).to(dev)
d_out = [torch.zeros(njobs * bench.TRACK_RES_DT.itemsize, dtype=torch.uint8, device=dev) for _ in range(nts)]
stream = torch.cuda.Stream(device=dev)
stream_b = torch.cuda.Stream(device=dev)
ctx = api.Context(0, stream=stream.cuda_stream)
ctx_b = api.Context(0, stream=stream_b.cuda_stream)
fbytes = nseq * W * H
L = capi.lib()
mk = lambda c: api.ImagePyramid(c, None, bench.SUBS, device_ptr=d_frames.data_ptr(), pitch=W, frame_stride=W * H, shape=(nseq, H, W))
pyrs = [mk(ctx_b), mk(ctx_b)]
def build(k):
    pyrs[k & 1].rebuild_d(d_frames.data_ptr() + (k % nts) * fbytes, W, W * H)
def walk(k):
    t = k % nts
    capi.check(ctx.handle, L.slam_hip_track_batch_d(ctx.handle, pyrs[k & 1].handle, C.c_void_p(d_jobs[t].data_ptr()), njobs, C.c_void_p(d_blob.data_ptr()), blob.size, C.c_void_p(d_out[t].data_ptr())))
# eager warm-up (grows queues)
for k in range(8):
    build(k); torch.cuda.synchronize(); walk(k); torch.cuda.synchronize()
ref = [d.clone() for d in d_out]
# capture 4 steps: pyramid k+1 beside walk k
g = torch.cuda.CUDAGraph()
with torch.cuda.stream(stream):
    build(0)  # pyramid of the first step, outside the graph
    torch.cuda.synchronize()
    g.capture_begin()
    for k in range(4):
        # fork: build k+1 on stream_b
        ev = torch.cuda.Event(); ev.record(stream); stream_b.wait_event(ev)
        build(k + 1)
        evb = torch.cuda.Event(); evb.record(stream_b)
        walk(k)
        stream.wait_event(evb)
    g.capture_end()
torch.cuda.synchronize()
for _ in range(5): g.replay()
torch.cuda.synchronize()
ok = all(torch.equal(a, b) for a, b in zip(d_out, ref))
n = 50
t0 = time.perf_counter()
for _ in range(n): g.replay()
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print("graph: %.4f ms per step, results equal eager: %s" % (dt / (4 * n) * 1e3, ok))
