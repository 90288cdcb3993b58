#!/usr/bin/env python3
"""Headline benchmark: NCC patch-match positions/s on the tracking hot path.

Workload (BASELINE.json configs[1], "C2"): 640x480 frames, 64 tracked features per frame, 21x21
templates with masks, 3-level pyramid {1,2,4}, stride 3, threshold 0.98 (Feature::track semantics);
a step = one pass over a batch of frames resident in HBM: pyramid build + coarse-to-fine search of
every feature of every frame.  Frames come from seeded synthetic sequences (frame t = frame 0
translated by a known integer shift); one process per GPU, frames sharded, no data-path collective.

  python bench.py --gpus N --steps K --warmup W        (N>1: launched by torch.distributed.run)

Prints ONE JSON line on rank 0.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

W, H = 640, 480
SUBS = [1.0, 2.0, 4.0]
NFEAT = 64
TSIZE = 21
STRIDE = 3
THRESHOLD = 0.98
SEQ_LEN = 32
HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak
HBM_COPY_GBS = 6290.0          # what a plain device-to-device copy reaches on this part (secondary denominator)
# Packed u8 dot (v_dot4_u32_u8): measured on gfx950 at ~4.5 cycles per wave64 instruction per SIMD
# (profiles/tools/valu_rate.hip), i.e. half the rate of a plain 32-bit VALU op:
# CUs x SIMDs x 64 lanes x 4 MAC / 4.5 clk x 2.4 GHz.
DOT4_PEAK_MACS = 256 * 4 * 64 * 4 / 4.5 * 2.4e9
TRAFFIC_FILE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r01_track_traffic.json")

TRACK_JOB_DT = np.dtype([("frame", "<i4"), ("projw", "<i4"), ("projh", "<i4"), ("_pad", "<i4"),
                         ("proj_off", "<i8"), ("mask_off", "<i8"), ("px", "<f4"), ("py", "<f4"),
                         ("threshold", "<f4"), ("stride", "<i4"),
                         ("amask_off", "<i8"), ("ax", "<i4"), ("ay", "<i4"), ("aw", "<i4"), ("ah", "<i4")])
TRACK_RES_DT = np.dtype([("x", "<i4"), ("y", "<i4"), ("response", "<f4"), ("dx", "<f4"), ("dy", "<f4"),
                         ("levels_run", "<i4"), ("status", "<i4")])


def level_cells_and_bytes():
    """Per level (coarse -> fine): score cells the lattice rule evaluates and algorithmic bytes
    (image tile under all candidate patches + template + mask + 20-byte result) for a feature
    whose windows lie inside the image."""
    out = []
    subs = SUBS
    for i in range(len(subs) - 1, -1, -1):
        sub = np.float32(subs[i])
        tw = TSIZE if sub == 1 else int(np.float32(TSIZE) / sub)
        if i == len(subs) - 1:
            rad = np.float32(20) / sub
        else:
            rad = np.float32(1.3 * float(np.float32(subs[i + 1])) / float(sub))
            if i == 0:
                rad = np.float32(rad * np.float32(STRIDE))
        win = int(np.floor(float(np.float32(2) * rad + np.float32(1)) + 0.5))
        s = 1 if i == 0 else STRIDE
        n1 = (win + s - 1) // s
        tile = (win + tw - 1) ** 2
        out.append(dict(level=i, tw=tw, win=win, cells=n1 * n1, bytes=tile + 2 * tw * tw + 20,
                        macs=2 * tw * tw * n1 * n1))
    return out


def build_workload(nframes, seed, rng):
    """Returns (frames[n,H,W] u8, jobs structured array, blob u8, truth[n*NFEAT,2])."""
    from slam_amd import synth
    nseq = (nframes + SEQ_LEN - 1) // SEQ_LEN
    frames = np.empty((nframes, H, W), np.uint8)
    jobs = np.zeros(nframes * NFEAT, TRACK_JOB_DT)
    truth = np.zeros((nframes * NFEAT, 2), np.int32)
    blob_parts = []
    blob_size = 0
    margin = TSIZE + 30
    pad = 8
    for s in range(nseq):
        big = synth.canvas(-pad, -pad, W + 2 * pad, H + 2 * pad, seed + s)
        f0 = big[pad:pad + H, pad:pad + W]
        pts = synth.feature_grid(W, H, NFEAT, margin, rng)
        offs = []
        for (cx, cy) in pts:
            t = synth.cut_template(f0, cx, cy, TSIZE, TSIZE)
            m = np.ones_like(t)
            k = int(rng.integers(0, TSIZE // 3 + 1))     # cut corners: a warped-quad-like mask
            for r in range(k):
                m[r, :k - r] = 0
                m[TSIZE - 1 - r, TSIZE - (k - r):] = 0
            offs.append((blob_size, blob_size + t.size))
            blob_parts += [t.reshape(-1), m.reshape(-1)]
            blob_size += 2 * t.size
        for t in range(SEQ_LEN):
            f = s * SEQ_LEN + t
            if f >= nframes:
                break
            u, v = (0, 0) if t == 0 else (int(rng.integers(-6, 7)), int(rng.integers(-6, 7)))
            frames[f] = big[pad + v:pad + v + H, pad + u:pad + u + W]
            for k, (cx, cy) in enumerate(pts):
                j = jobs[f * NFEAT + k]
                j["frame"] = f
                j["projw"] = j["projh"] = TSIZE
                j["proj_off"], j["mask_off"] = offs[k]
                tx, ty = cx - u, cy - v
                j["px"] = tx + int(rng.integers(-3, 4)) + float(rng.random()) * 0.9
                j["py"] = ty + int(rng.integers(-3, 4)) + float(rng.random()) * 0.9
                j["threshold"] = THRESHOLD
                j["stride"] = STRIDE
                truth[f * NFEAT + k] = (tx, ty)
    return frames, jobs, np.concatenate(blob_parts), truth


def oracle_run(frames, jobs, blob, nfr):
    """The CPU oracle on the first nfr frames; returns (results, seconds, positions)."""
    from oracle import pyoracle as O
    O.build()
    lc = level_cells_and_bytes()
    res = []
    t0 = time.perf_counter()
    for f in range(nfr):
        levels = O.pyramid(frames[f], SUBS)
        for k in range(NFEAT):
            j = jobs[f * NFEAT + k]
            n = TSIZE * TSIZE
            proj = blob[j["proj_off"]:j["proj_off"] + n].reshape(TSIZE, TSIZE)
            mask = blob[j["mask_off"]:j["mask_off"] + n].reshape(TSIZE, TSIZE)
            res.append(O.track(levels, SUBS, proj, mask, float(j["px"]), float(j["py"]), THRESHOLD, STRIDE))
    dt = time.perf_counter() - t0
    pos = sum(sum(l["cells"] for l in lc[:r["levels_run"]]) for r in res)
    return res, dt, pos


def _oracle_worker(args):
    """One process of the all-cores CPU baseline: tracks a slice of frames with the oracle."""
    frames, jobs, blob = args
    _, dt, pos = oracle_run(frames, jobs, blob, len(frames))
    return dt, pos


def oracle_run_parallel(frames, jobs, blob, nfr, nproc):
    """The same oracle pass split over nproc worker processes (spawned: they never touch the GPU).
    Returns (wall seconds of the parallel section, positions)."""
    import multiprocessing as mp
    parts = []
    for k in range(nproc):
        lo, hi = nfr * k // nproc, nfr * (k + 1) // nproc
        if hi > lo:
            jj = jobs[lo * NFEAT:hi * NFEAT].copy()
            jj["frame"] -= lo
            parts.append((frames[lo:hi], jj, blob))
    with mp.get_context("spawn").Pool(len(parts)) as pool:
        pool.map(_oracle_worker, [(p[0][:1], p[1][:NFEAT], p[2]) for p in parts])      # start-up and page-in, untimed
        t0 = time.perf_counter()
        res = pool.map(_oracle_worker, parts)
        wall = time.perf_counter() - t0
    return wall, sum(r[1] for r in res)


def fast_leg(api, capi, ctx, stream, dev, args):
    """FAST-9-16 + NMS px/s on C4 (BASELINE.json configs[3]): 64 frames 1920x1080 resident in HBM, threshold 10
    (SLAM::fastThreshold's initial value, SLAM.hpp:177); frame 0's corner list is compared with the oracle's first."""
    import torch
    from slam_amd import synth
    w, h, nfr, cap, thr = 1920, 1080, 64, 262144, 10
    base = np.stack([synth.frame(w, h, 0x51A40004 + i) for i in range(4)])
    d = torch.from_numpy(base).to(dev).repeat(nfr // 4, 1, 1).contiguous()
    L = capi.lib()
    sb = L.slam_hip_fast_scratch_bytes(w, h, nfr, cap)
    d_scr = torch.zeros(sb, dtype=torch.uint8, device=dev)
    d_out = torch.zeros(nfr * cap * 12, dtype=torch.uint8, device=dev)
    d_cnt = torch.zeros(nfr, dtype=torch.int32, device=dev)

    def go():
        capi.check(ctx.handle, L.slam_hip_fast_detect_d(ctx.handle, C.c_void_p(d.data_ptr()), w, h, w, w * h, nfr, thr, 1,
                                                        C.c_void_p(d_out.data_ptr()), cap, C.c_void_p(d_cnt.data_ptr()),
                                                        C.c_void_p(d_scr.data_ptr()), sb))
    go()
    torch.cuda.synchronize()
    cnt = d_cnt.cpu().numpy()
    parity = None
    if not args.no_cpu:
        from oracle import pyoracle as O
        ref = O.fast(base[0], thr, True)
        got = d_out[:cap * 12].cpu().numpy().view(np.float32).reshape(cap, 3)[:min(int(cnt[0]), cap)]
        if len(ref) != cnt[0] or not np.array_equal(got, ref[:cap]):
            raise SystemExit("PARITY FAILURE: FAST corner list of frame 0 differs from the oracle's")
        parity = "frame 0: %d corners, list identical to the oracle's" % len(ref)
    steps = max(5, args.steps)
    for _ in range(3):
        go()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    for a, b in evs:
        a.record(stream); go(); b.record(stream)
    torch.cuda.synchronize()
    ms = float(np.mean([a.elapsed_time(b) for a, b in evs]))
    alg = w * h * nfr + 12 * int(np.minimum(cnt, cap).sum())
    return {"workload": "C4: %d frames %dx%d, FAST-9-16 threshold %d + non-max suppression" % (nfr, w, h, thr),
            "value": w * h * nfr / (ms * 1e-3), "unit": "px/s", "ms": ms, "corners_per_frame": float(cnt.mean()),
            "roofline": {"bound": "hbm", "kernel": "fast_tile_kernel+fast_scan_kernel+fast_scatter_kernel", "achieved": alg / (ms * 1e-3) / 1e9,
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": alg / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                         "algorithmic_bytes_per_launch": alg},
            "parity": parity}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--frames", type=int, default=1024, help="frames per GPU per step")
    ap.add_argument("--cpu-frames", type=int, default=1024, help="frames of the CPU-baseline sample")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="repeat the CPU sample until this much CPU time")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-fast", action="store_true", help="skip the FAST (C4) leg")
    ap.add_argument("--cpu-procs", type=int, default=16,
                    help="worker processes of the additional all-cores CPU figure (0 = skip it)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL)")
    ap.add_argument("--rehearse-one-device", action="store_true",
                    help="diagnostic: every rank uses cuda:0 (rehearsal of the N > 1 control flow on a one-GPU box, with --backend gloo)")
    ap.add_argument("--with-h2d", action="store_true",
                    help="diagnostic: each step also uploads the frames from pinned host memory (PCIe-inclusive rate)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from slam_amd import api, capi, parallel

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run --nproc-per-node N" % (args.gpus, world))
    if args.rehearse_one_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    rdev = dev if args.backend == "nccl" else torch.device("cpu")     # where the two reduced scalars live
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(args.backend)

    rng = np.random.default_rng(0x51A40002 + 7919 * rank)
    frames, jobs, blob, truth = build_workload(args.frames, 0x51A40002 + 1000 * rank, rng)
    nframes, njobs = args.frames, args.frames * NFEAT

    # inputs resident in HBM before the timed region
    d_frames = torch.from_numpy(frames).to(dev)
    d_jobs = torch.from_numpy(jobs.view(np.uint8).reshape(-1)).to(dev)
    d_blob = torch.from_numpy(blob).to(dev)
    d_out = torch.zeros(njobs * TRACK_RES_DT.itemsize, dtype=torch.uint8, device=dev)
    stream = torch.cuda.Stream(device=dev)      # a real (non-null) stream shared by the kernels and the events
    torch.cuda.set_stream(stream)
    ctx = api.Context(local_rank, stream=stream.cuda_stream)
    pyr = api.ImagePyramid(ctx, None, SUBS, device_ptr=d_frames.data_ptr(), pitch=W, frame_stride=W * H,
                           shape=(nframes, H, W))
    L = capi.lib()

    h_frames = torch.from_numpy(frames).pin_memory() if args.with_h2d else None

    def step(ev=None):
        if h_frames is not None:
            d_frames.copy_(h_frames, non_blocking=True)
        if ev is not None:
            ev[2].record(stream)
        pyr.rebuild_d(d_frames.data_ptr(), W, W * H)
        if ev is not None:
            ev[0].record(stream)
        capi.check(ctx.handle, L.slam_hip_track_batch_d(ctx.handle, pyr.handle, C.c_void_p(d_jobs.data_ptr()), njobs,
                                                        C.c_void_p(d_blob.data_ptr()), blob.size,
                                                        C.c_void_p(d_out.data_ptr())))
        if ev is not None:
            ev[1].record(stream)

    # ---- parity gate: results must equal the oracle's before any timing counts ------------------
    step()
    torch.cuda.synchronize()
    res = d_out.cpu().numpy().view(TRACK_RES_DT)
    assert (res["status"] == 0).all(), "device rejected jobs"
    lc = level_cells_and_bytes()
    cells = np.array([0] + list(np.cumsum([l["cells"] for l in lc])))
    lbytes = np.array([0] + list(np.cumsum([l["bytes"] for l in lc])))
    lmacs = np.array([0] + list(np.cumsum([l["macs"] for l in lc])))
    positions = int(cells[res["levels_run"]].sum())
    alg_bytes = int(lbytes[res["levels_run"]].sum())
    macs = int(lmacs[res["levels_run"]].sum())
    tracked = float(np.mean((res["x"] == truth[:, 0]) & (res["y"] == truth[:, 1]) & (res["response"] > THRESHOLD)))

    cpu = None
    if rank == 0 and not args.no_cpu:
        nfr = min(args.cpu_frames, nframes)
        ores, cpu_s, cpu_pos = oracle_run(frames, jobs, blob, nfr)
        for i, r in enumerate(ores):
            g = res[i]
            same = (g["x"], g["y"], g["levels_run"]) == (r["x"], r["y"], r["levels_run"]) and \
                np.float32(g["response"]).view(np.uint32) == np.float32(r["response"]).view(np.uint32)
            if not same:
                raise SystemExit("PARITY FAILURE job %d: gpu %s oracle %s" % (i, g, r))
        passes = 1
        while cpu_s < args.cpu_seconds and passes < 16:       # same sample again until the CPU time is long enough
            _, dt, pos = oracle_run(frames, jobs, blob, nfr)
            cpu_s += dt; cpu_pos += pos; passes += 1
        cpu = {"value": cpu_pos / cpu_s, "unit": "positions/s", "cores": 1, "kind": "port",
               "sample": "CPU oracle (pyramid + Feature::track) on the first %d frames x %d features of the same batch, "
                         "%d passes, single thread, %.1f s in all; results bit-identical to the GPU's"
                         % (nfr, NFEAT, passes, cpu_s)}
        if args.cpu_procs > 1:     # secondary figure: the same port on the host cores this job may use
            wall, pos = oracle_run_parallel(frames, jobs, blob, nfr, args.cpu_procs)
            cpu["all_cores"] = {"value": pos / wall, "unit": "positions/s", "cores": args.cpu_procs,
                                "sample": "same frames split over %d processes, %.1f s wall" % (args.cpu_procs, wall)}

    # ---- timing --------------------------------------------------------------------------------
    for _ in range(args.warmup):
        step()
    evs = [tuple(torch.cuda.Event(enable_timing=True) for _ in range(3)) for _ in range(args.steps)]
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(evs[k])
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0

    kern_ms = float(np.mean([e[0].elapsed_time(e[1]) for e in evs]))
    pyr_ms = float(np.mean([e[2].elapsed_time(e[0]) for e in evs]))
    pyr_bytes = nframes * (W * H + sum(pyr.level(i).w * pyr.level(i).h for i in range(1, len(SUBS))))
    tot_pos = float(positions)
    elapsed, tot_pos = parallel.reduce_max_sum(elapsed, tot_pos, rdev)
    fast = fast_leg(api, capi, ctx, stream, dev, args) if (rank == 0 and not args.no_fast) else None

    if rank == 0:
        traffic = None          # HBM bytes per launch from the committed rocprofv3 --pmc passes of this command
        if os.path.exists(TRAFFIC_FILE):
            with open(TRAFFIC_FILE) as fh:
                tf = json.load(fh)
            if tf.get("frames_per_gpu") == nframes:
                traffic = tf["track_kernel"]["hbm_bytes_per_launch"]
        ach = alg_bytes / (kern_ms * 1e-3) / 1e9
        line = {
            "metric": "NCC patch-match positions/sec (pyramid build + coarse-to-fine template search)",
            "value": tot_pos * args.steps / elapsed,
            "unit": "positions/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8", "data": "synthetic",
            "config": {"workload": "C2: 640x480 sequences (T=%d), %d features/frame, %dx%d masked templates, "
                                   "pyramid {1,2,4}, stride 3, threshold 0.98; %d frames resident per GPU per step"
                                   % (SEQ_LEN, NFEAT, TSIZE, TSIZE, nframes),
                       "frames_per_gpu": nframes, "features_per_frame": NFEAT, "positions_per_step_per_gpu": positions,
                       "frames_per_s": nframes * world * args.steps / elapsed, "tracked_fraction": tracked, "h2d_in_step": bool(args.with_h2d),
                       "parallelism": "frames sharded over %d GPU(s), no collective on the data path" % world},
            "roofline": {"bound": "hbm", "kernel": "track_kernel", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": ach / HBM_PEAK_GBS, "frac_of_measured_copy_rate": ach / HBM_COPY_GBS, "traffic": traffic,
                         "algorithmic_bytes_per_launch": alg_bytes, "kernel_ms": kern_ms,
                         "u8_mac_per_s": macs / (kern_ms * 1e-3), "frac_of_packed_dot_peak": macs / (kern_ms * 1e-3) / DOT4_PEAK_MACS},
            "cpu_baseline": cpu,
            # the other two kernels BASELINE.json's metric names, measured in the same run (outside the timed steps)
            "pyramid": {"kernel": "pyr_build_kernel", "ms": pyr_ms, "achieved": pyr_bytes / (pyr_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS,
                        "unit": "GB/s", "frac": pyr_bytes / (pyr_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                        "frac_of_measured_copy_rate": pyr_bytes / (pyr_ms * 1e-3) / 1e9 / HBM_COPY_GBS,
                        "algorithmic_bytes_per_launch": pyr_bytes},
            "fast": fast,
        }
        print(json.dumps(line))
    pyr.close()
    ctx.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
