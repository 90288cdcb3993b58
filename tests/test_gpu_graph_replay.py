"""The `_d` entry points only enqueue work on their context's stream, so a caller may capture them in a HIP graph and
replay it.  A replay runs the same kernels with the same arguments as the capture: whatever state a launch needs (the
counters of the queue between the lean and the general kernel) must be left in place by the launch before it, not by
host code that a replay never runs.  One captured walk (an odd number of launches per graph) with jobs on both kernels
is replayed several times against the eager results; the same for a batch of single searches."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

from conftest import DEFAULT_KERNEL_SPLIT

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

pytestmark = pytest.mark.gpu


def _walks(nseq, seed):
    """Bench-shaped walks, every fifth pushed to the image corner (the lean kernel leaves those to the general one)."""
    import bench
    frames, jobs, blob, _ = bench.build_time_steps(nseq, 1, seed, np.random.default_rng(seed))
    frames, jobs = frames[0], jobs[0].copy()
    edge = np.arange(len(jobs)) % 5 == 0
    jobs["px"][edge] = 3.0 + (np.arange(len(jobs))[edge] % 7)
    jobs["py"][edge] = 2.0 + (np.arange(len(jobs))[edge] % 5)
    return frames, jobs, blob


def _replay_against_eager(ctx, stream, call, d_out, nbytes, replays=4):
    import torch
    from slam_amd import capi
    L = capi.lib()
    junk = np.full(nbytes, 0xAB, np.uint8)
    capi.check(ctx.handle, call())                       # eager: also grows the context's queue before the capture
    ctx.synchronize()
    want = d_out.get().copy()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=stream):
        capi.check(ctx.handle, call())
    for rep in range(replays):
        capi.check(ctx.handle, L.slam_hip_memcpy_h2d(ctx.handle, d_out.ptr, junk.ctypes.data, nbytes))
        g.replay()
        torch.cuda.synchronize()
        got = d_out.get()
        assert np.array_equal(got, want), (rep, np.flatnonzero(got != want)[:8])
    # and the eager path still works after the replays (the queue's counters are where a launch expects them)
    capi.check(ctx.handle, L.slam_hip_memcpy_h2d(ctx.handle, d_out.ptr, junk.ctypes.data, nbytes))
    capi.check(ctx.handle, call())
    ctx.synchronize()
    assert np.array_equal(d_out.get(), want)
    return want


@pytest.mark.parametrize("entry", ["slam_hip_track_batch_d", "slam_hip_track_batch_noalloc_d"])
def test_one_captured_walk_replays_like_the_eager_call(entry):
    import torch
    import bench
    from slam_amd import api, capi
    L = capi.lib()
    stream = torch.cuda.Stream(device=torch.device("cuda", 0))
    with api.Context(0, stream=stream.cuda_stream) as ctx:
        frames, jobs, blob = _walks(12, 931)
        pyr = api.ImagePyramid(ctx, frames, bench.SUBS)
        d_jobs = api.DeviceBuffer(ctx, host=jobs.view(np.uint8).reshape(-1))
        d_blob = api.DeviceBuffer(ctx, host=blob)
        nbytes = len(jobs) * bench.TRACK_RES_DT.itemsize
        d_out = api.DeviceBuffer(ctx, nbytes=nbytes, fill=0xAB)
        try:
            call = lambda: getattr(L, entry)(ctx.handle, pyr.handle, d_jobs.ptr, len(jobs), d_blob.ptr, blob.size, d_out.ptr)
            want = _replay_against_eager(ctx, stream, call, d_out, nbytes).view(bench.TRACK_RES_DT)
            assert (want["status"] == 0).all() and (want["levels_run"] >= 1).all()
            deferred = C.c_int32(-1)
            capi.check(ctx.handle, L.slam_hip_debug_walk_deferred(ctx.handle, C.byref(deferred)))
            if DEFAULT_KERNEL_SPLIT:
                assert 0 < deferred.value < len(jobs)    # both kernels had work in every replay
        finally:
            d_jobs.free(); d_blob.free(); d_out.free(); pyr.close()


def test_one_captured_walk_in_bands_replays_like_the_eager_call(monkeypatch):
    """The second walk build with the last level in bands of rows (forced for this batch size): band kernel, finish kernel
    and the general kernel over the queue, captured once and replayed; the band records live in a buffer the eager call
    before the capture has grown."""
    import torch
    import bench
    from slam_amd import api, capi
    monkeypatch.setenv("SLAM_HIP_WALK_BANDS", "4")
    L = capi.lib()
    stream = torch.cuda.Stream(device=torch.device("cuda", 0))
    with api.Context(0, stream=stream.cuda_stream) as ctx:
        capi.check(ctx.handle, L.slam_hip_set_walk_chunks(ctx.handle, 1))
        frames, jobs, blob = _walks(6, 932)
        pyr = api.ImagePyramid(ctx, frames, bench.SUBS)
        d_jobs = api.DeviceBuffer(ctx, host=jobs.view(np.uint8).reshape(-1))
        d_blob = api.DeviceBuffer(ctx, host=blob)
        nbytes = len(jobs) * bench.TRACK_RES_DT.itemsize
        d_out = api.DeviceBuffer(ctx, nbytes=nbytes, fill=0xAB)
        try:
            call = lambda: L.slam_hip_track_batch_d(ctx.handle, pyr.handle, d_jobs.ptr, len(jobs), d_blob.ptr, blob.size, d_out.ptr)
            want = _replay_against_eager(ctx, stream, call, d_out, nbytes).view(bench.TRACK_RES_DT)
            assert (want["status"] == 0).all() and (want["levels_run"] >= 1).all()
            capi.check(ctx.handle, L.slam_hip_set_walk_chunks(ctx.handle, 0))          # the plain build gives the same records
            capi.check(ctx.handle, call())
            ctx.synchronize()
            assert np.array_equal(d_out.get().view(bench.TRACK_RES_DT), want)
        finally:
            d_jobs.free(); d_blob.free(); d_out.free(); pyr.close()


def test_one_captured_batch_of_single_searches_replays_like_the_eager_call():
    import torch
    import bench
    from slam_amd import api, capi, synth
    L = capi.lib()
    stream = torch.cuda.Stream(device=torch.device("cuda", 0))
    rng = np.random.default_rng(77)
    w, h = 320, 240
    frame = synth.frame(w, h, 12)
    with api.Context(0, stream=stream.cuda_stream) as ctx:
        pyr = api.ImagePyramid(ctx, frame, [1.0])
        blob = api.Blob()
        jobs = np.zeros(96, bench.MATCH_JOB_DT)
        for i in range(len(jobs)):
            cx, cy = int(rng.integers(40, w - 40)), int(rng.integers(40, h - 40))
            if i % 4 == 0:
                cx = 2                                   # a window over the border: general kernel
            patch = synth.cut_template(frame, max(cx, 8), cy, 11, 11)
            mask = (rng.random((11, 11)) < 0.8).astype(np.uint8)
            jobs[i]["frame"] = 0; jobs[i]["level"] = 0; jobs[i]["pw"] = 11; jobs[i]["ph"] = 11
            jobs[i]["patch_off"] = blob.add(patch); jobs[i]["pmask_off"] = blob.add(mask) if i % 2 else -1
            jobs[i]["ax"] = cx - 8; jobs[i]["ay"] = cy - 8; jobs[i]["aw"] = 17; jobs[i]["ah"] = 17
            jobs[i]["amask_off"] = -1; jobs[i]["stride"] = 1; jobs[i]["use_exact_area_mask"] = 0; jobs[i]["score_off"] = -1
        bb = blob.bytes()
        d_jobs = api.DeviceBuffer(ctx, host=jobs.view(np.uint8).reshape(-1))
        d_blob = api.DeviceBuffer(ctx, host=bb)
        nbytes = len(jobs) * bench.MATCH_RES_DT.itemsize
        d_out = api.DeviceBuffer(ctx, nbytes=nbytes, fill=0xAB)
        try:
            call = lambda: L.slam_hip_match_batch_d(ctx.handle, pyr.handle, d_jobs.ptr, len(jobs), d_blob.ptr, bb.size,
                                                    None, 0, d_out.ptr)
            want = _replay_against_eager(ctx, stream, call, d_out, nbytes).view(bench.MATCH_RES_DT)
            assert (want["status"] == 0).all()
        finally:
            d_jobs.free(); d_blob.free(); d_out.free(); pyr.close()
