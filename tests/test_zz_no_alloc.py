"""slam_hip_track_batch_noalloc_d (the walk entry point that allocates nothing: its deferred-job queue is a fixed array
inside the library) must give exactly what slam_hip_track_batch_d gives, call after call on one GPU - also when the
second call is smaller than the first and the array still holds the first call's queue."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

from conftest import DEFAULT_KERNEL_SPLIT

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

pytestmark = pytest.mark.gpu


def _workload(nseq, seed):
    """Bench-shaped walks plus some the lean kernel must leave to the general one (windows over the image border)."""
    import bench
    frames, jobs, blob, _ = bench.build_time_steps(nseq, 1, seed, np.random.default_rng(seed))
    frames, jobs = frames[0], jobs[0].copy()
    edge = np.arange(len(jobs)) % 5 == 0                 # every fifth prediction is pushed to the image corner
    jobs["px"][edge] = 3.0 + (np.arange(len(jobs))[edge] % 7)
    jobs["py"][edge] = 2.0 + (np.arange(len(jobs))[edge] % 5)
    return frames, jobs, blob


def _run(entry, ctx, pyr, jobs, blob):
    import bench
    from slam_amd import api, capi
    d_jobs = api.DeviceBuffer(ctx, host=jobs.view(np.uint8).reshape(-1))
    d_blob = api.DeviceBuffer(ctx, host=blob)
    d_out = api.DeviceBuffer(ctx, nbytes=len(jobs) * bench.TRACK_RES_DT.itemsize, fill=0xAB)
    try:
        capi.check(ctx.handle, entry(ctx.handle, pyr.handle, d_jobs.ptr, len(jobs), d_blob.ptr, blob.size, d_out.ptr))
        ctx.synchronize()
        return d_out.get().view(bench.TRACK_RES_DT).copy()
    finally:
        d_jobs.free(); d_blob.free(); d_out.free()


def test_noalloc_walk_equals_the_allocating_entry_point_twice_in_a_row():
    import bench
    from slam_amd import api, capi
    L = capi.lib()
    with api.Context(0) as ctx:
        for nseq, seed in ((24, 901), (5, 902)):        # the second call is the smaller one
            frames, jobs, blob = _workload(nseq, seed)
            pyr = api.ImagePyramid(ctx, frames, bench.SUBS)
            want = _run(L.slam_hip_track_batch_d, ctx, pyr, jobs, blob)
            deferred = C.c_int32(-1)
            capi.check(ctx.handle, L.slam_hip_debug_walk_deferred(ctx.handle, C.byref(deferred)))
            got = _run(L.slam_hip_track_batch_noalloc_d, ctx, pyr, jobs, blob)
            assert (want["status"] == 0).all() and (want["levels_run"] >= 1).all()
            if DEFAULT_KERNEL_SPLIT:   # cross-check runs skip the lean kernel
                assert 0 < deferred.value < len(jobs)    # both kernels had work: the queue was really used
            assert np.array_equal(got, want), (nseq, np.flatnonzero(got != want)[:8])
            pyr.close()


def test_noalloc_walk_leaves_the_free_device_memory_unchanged():
    """hipMemGetInfo before and after a run of calls (through torch, which holds no cached block of its own here): the
    entry point allocates nothing, whatever the batch size - also a first batch larger than any the context has seen,
    which makes the allocating entry point grow its queue."""
    import torch
    import bench
    from slam_amd import api, capi
    L = capi.lib()
    with api.Context(0) as ctx:
        frames, jobs, blob = _workload(40, 905)
        pyr = api.ImagePyramid(ctx, frames, bench.SUBS)
        d_jobs = api.DeviceBuffer(ctx, host=jobs.view(np.uint8).reshape(-1))
        d_blob = api.DeviceBuffer(ctx, host=blob)
        d_out = api.DeviceBuffer(ctx, nbytes=len(jobs) * bench.TRACK_RES_DT.itemsize, fill=0)
        try:
            ctx.synchronize()
            torch.cuda.synchronize()
            free0, _ = torch.cuda.mem_get_info(0)
            for n in (64, len(jobs), 7):
                capi.check(ctx.handle, L.slam_hip_track_batch_noalloc_d(ctx.handle, pyr.handle, d_jobs.ptr, n, d_blob.ptr, blob.size, d_out.ptr))
            ctx.synchronize()
            free1, _ = torch.cuda.mem_get_info(0)
            assert free1 == free0, (free0, free1)
        finally:
            d_jobs.free(); d_blob.free(); d_out.free(); pyr.close()


def test_noalloc_walk_rejects_batches_beyond_its_queue():
    import bench
    from slam_amd import api, capi
    L = capi.lib()
    with api.Context(0) as ctx:
        frames, jobs, blob = _workload(1, 903)
        pyr = api.ImagePyramid(ctx, frames, bench.SUBS)
        rc = L.slam_hip_track_batch_noalloc_d(ctx.handle, pyr.handle, C.c_void_p(16), (1 << 20) + 1, C.c_void_p(16), 1, C.c_void_p(16))
        assert rc == capi.ECAP
        assert b"at most" in L.slam_hip_last_error_string(ctx.handle)
        pyr.close()
