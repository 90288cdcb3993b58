#!/usr/bin/env python3
"""Headline benchmark: NCC patch-match positions/s on the tracking hot path.

Workload (BASELINE.json configs[1] "C2" at N = 1, configs[4] "C5" at N = 8): independent 640x480
sequences, 64 tracked features per frame, 21x21 templates with masks, 3-level pyramid {1,2,4},
stride 3, threshold 0.98 (Feature::track semantics).  Every GPU owns `--sequences-per-gpu`
sequences (default 512: 8 GPUs = the 4096 sequences of C5); a step = one time-step of all of a
GPU's sequences, i.e. one frame per sequence resident in HBM: pyramid build + coarse-to-fine
search of every feature of every frame.  `--time-steps` time-steps of every sequence are resident
and the steps cycle through them, so consecutive steps never touch the same frames.  One process
per GPU, sequences sharded, no data-path collective.

  python bench.py --gpus N --steps K --warmup W

With N > 1 and no torch.distributed environment the command starts its own N rank processes
(torch.distributed.run, rendezvous on 127.0.0.1) before anything touches a GPU; under
torch.distributed.run it is a rank.  Rank 0 prints ONE JSON line.
"""
import argparse
import ctypes as C
import json
import os
import socket
import subprocess
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
HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak
HBM_COPY_GBS = 6290.0          # what a plain device-to-device copy reaches on this part (secondary denominator)
# Packed u8 dot (v_dot4_u32_u8) peaks, in u8 multiply-accumulates per second over 256 CUs x 4 SIMDs x 64 lanes x 4 MAC:
#  - measured: with 2 to 8 waves per SIMD issuing nothing else, one wave instruction costs the wall time of 4.25-4.8
#    cycles of a 2.4 GHz clock per SIMD (profiles/r02_valu_rate.txt: 2.6-4.5 shader cycles at the 1.5-2.3 GHz the
#    chip holds under that load); 4.25 is used;
#  - nominal: 2 cycles per wave64 instruction on a SIMD-32 at 2.4 GHz, which no instruction but v_add/v_and/v_lshl
#    came near in that measurement.
DOT4_PEAK_MEASURED = 256 * 4 * 64 * 4 / 4.25 * 2.4e9
DOT4_PEAK_NOMINAL = 256 * 4 * 64 * 4 / 2.0 * 2.4e9
TRAFFIC_FILE = os.path.join(ROOT, "profiles", "r03_traffic.json")
LDS_PEAK_BPS = 150e12           # ds_read_b64 / b128 with every CU streaming (256 B per clock per CU at 2.4 GHz: 157 TB/s nominal)

TRACK_JOB_DT = np.dtype([("frame", "<i4"), ("projw", "<i4"), ("projh", "<i4"), ("flags", "<i4"),
                         ("proj_off", "<i8"), ("mask_off", "<i8"), ("px", "<f4"), ("py", "<f4"),
                         ("threshold", "<f4"), ("stride", "<i4"),
                         ("amask_off", "<i8"), ("ax", "<i4"), ("ay", "<i4"), ("aw", "<i4"), ("ah", "<i4")])
TRACK_RES_DT = np.dtype([("x", "<i4"), ("y", "<i4"), ("response", "<f4"), ("dx", "<f4"), ("dy", "<f4"),
                         ("levels_run", "<i4"), ("status", "<i4")])
MATCH_JOB_DT = np.dtype([("frame", "<i4"), ("level", "<i4"), ("pw", "<i4"), ("ph", "<i4"),
                         ("patch_off", "<i8"), ("pmask_off", "<i8"), ("ax", "<i4"), ("ay", "<i4"), ("aw", "<i4"), ("ah", "<i4"),
                         ("amask_off", "<i8"), ("stride", "<i4"), ("use_exact_area_mask", "<i4"), ("score_off", "<i8")])
MATCH_RES_DT = np.dtype([("x", "<i4"), ("y", "<i4"), ("response", "<f4"), ("dx", "<f4"), ("dy", "<f4"), ("status", "<i4")])
LINE_JOB_DT = np.dtype([("desc_off", "<i8"), ("amask_off", "<i8"), ("frame", "<i4"), ("dw", "<i4"), ("dh", "<i4"),
                        ("ax", "<i4"), ("ay", "<i4"), ("aw", "<i4"), ("ah", "<i4"), ("threshold", "<f4"),
                        ("stride", "<i4"), ("reserved", "<i4")])


# ---------------------------------------------------------------------------------------------------
# launcher
# ---------------------------------------------------------------------------------------------------
def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--sequences-per-gpu", type=int, default=512,
                    help="independent sequences a GPU owns; a step processes one frame of each (C5: 512 x 8 GPUs)")
    ap.add_argument("--time-steps", type=int, default=4, help="time-steps of every sequence resident in HBM; steps cycle through them")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="repeat the CPU sample until this much CPU time")
    ap.add_argument("--no-overlap", action="store_true",
                    help="build each time-step's pyramid on the walk's own stream, right before its walk (default: the pyramid "
                         "of time-step k+1 is built on a second stream while time-step k is walked; two pyramids, double-buffered)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the oracle: no parity gates, no cpu_baseline")
    ap.add_argument("--no-side", action="store_true", help="skip the extra legs (C1, C3, line walk, FAST)")
    ap.add_argument("--no-fast", action="store_true", help="skip the FAST (C4) leg")
    ap.add_argument("--legs", default="fast,c1,c3,prod,line",
                    help="the extra legs to run, of fast,c1,c3,prod,line (profiling: the general walk kernel has work in the prod leg only)")
    ap.add_argument("--c3-walk-lds", type=int, default=10240, help="LDS bytes per walk of the lean kernel in the C3 leg")
    ap.add_argument("--c3-chunks", action="store_true", help="C3 leg through the lean kernel's build with template chunks")
    ap.add_argument("--prod-walk-lds", type=int, default=12800,
                    help="LDS bytes per walk of the lean kernel's chunked build in the prod leg (0: general kernel only)")
    ap.add_argument("--no-fast-sparse", action="store_true",
                    help="FAST leg at threshold 10 only (profiling: keeps one kind of launch per kernel name)")
    ap.add_argument("--repeats", type=int, default=0,
                    help="timed blocks of --steps steps (0 = as many as make 0.25 s of GPU time, at least 5); the line reports the median block")
    ap.add_argument("--cpu-procs", type=int, default=-1,
                    help="worker processes of the additional all-cores CPU figure (0 = skip it; default: the host cores this "
                         "process may use, capped by OMP_NUM_THREADS, which a shared host sets to the job's CPU share)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL)")
    ap.add_argument("--rehearse-one-device", action="store_true",
                    help="diagnostic: every rank uses cuda:0 (rehearsal of the N > 1 control flow on a one-GPU box, with --backend gloo)")
    ap.add_argument("--with-h2d", action="store_true",
                    help="diagnostic: each step also uploads its frames from pinned host memory (PCIe-inclusive rate)")
    return ap.parse_args(argv)


def rank_launch_command(args, argv, port):
    """The command line that runs this file as N ranks of one node (one process per GPU)."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % args.gpus,
            "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


def launch_ranks(args, argv):
    """`python bench.py --gpus N` outside torch.distributed.run: start the N ranks as child processes.  This process
    never touches a GPU (torch is not even imported here); the children's output passes through."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(rank_launch_command(args, argv, port), env=env)


# ---------------------------------------------------------------------------------------------------
# C2 / C5 workload
# ---------------------------------------------------------------------------------------------------
def level_cells_and_bytes(tsize=TSIZE, subs=SUBS, stride=STRIDE):
    """Per level (coarse -> fine): score cells the lattice rule evaluates and algorithmic bytes
    (image tile under all candidate patches + template + mask + 20-byte result) for a feature
    whose windows lie inside the image."""
    out = []
    for i in range(len(subs) - 1, -1, -1):
        sub = np.float32(subs[i])
        tw = tsize if sub == 1 else int(np.float32(tsize) / sub)
        if i == len(subs) - 1:
            rad = np.float32(20) / sub
        else:
            rad = np.float32(1.3 * float(np.float32(subs[i + 1])) / float(sub))
            if i == 0:
                rad = np.float32(rad * np.float32(stride))
        win = int(np.floor(float(np.float32(2) * rad + np.float32(1)) + 0.5))
        s = 1 if i == 0 else stride
        n1 = (win + s - 1) // s
        tile = (win + tw - 1) ** 2
        out.append(dict(level=i, tw=tw, win=win, cells=n1 * n1, bytes=tile + 2 * tw * tw + 20,
                        macs=2 * tw * tw * n1 * n1))
    return out


def build_time_steps(nseq, nsteps, seed, rng):
    """nseq independent sequences, nsteps time-steps of each.  Returns (frames[nsteps, nseq, H, W] u8, jobs: one
    structured array per time-step (frame = sequence index within the time-step), blob u8, truth[nsteps][nseq*NFEAT, 2]).
    A sequence is a window sliding over its own procedural canvas (band-limited value noise, synth.canvas_smooth): the frame at time t is frame 0 translated by an
    integer (u, v), |u|, |v| <= 6; templates are cut from frame 0 with warped-quad-like masks (cut corners); the
    prediction handed to the search is the truth plus an error of up to 3.9 pixels per axis."""
    from slam_amd import synth
    frames = np.empty((nsteps, nseq, H, W), np.uint8)
    jobs = [np.zeros(nseq * NFEAT, TRACK_JOB_DT) for _ in range(nsteps)]
    truth = [np.zeros((nseq * NFEAT, 2), np.int32) for _ in range(nsteps)]
    blob_parts = []
    blob_size = 0
    margin = TSIZE + 30
    pad = 8
    for s in range(nseq):
        big = synth.canvas_smooth(-pad, -pad, W + 2 * pad, H + 2 * pad, seed + s)
        f0 = big[pad:pad + H, pad:pad + W]
        pts = np.array(synth.feature_grid(W, H, NFEAT, margin, rng), np.int64)
        offs = np.zeros((NFEAT, 2), np.int64)
        for k, (cx, cy) in enumerate(pts):
            t = synth.cut_template(f0, int(cx), int(cy), TSIZE, TSIZE)
            m = np.ones_like(t)
            c = int(rng.integers(0, TSIZE // 3 + 1))     # cut corners: a warped-quad-like mask
            for r in range(c):
                m[r, :c - r] = 0
                m[TSIZE - 1 - r, TSIZE - (c - r):] = 0
            offs[k] = (blob_size, blob_size + t.size)
            blob_parts += [t.reshape(-1), m.reshape(-1)]
            blob_size += 2 * t.size
        for t in range(nsteps):
            u, v = int(rng.integers(-6, 7)), int(rng.integers(-6, 7))
            frames[t, s] = big[pad + v:pad + v + H, pad + u:pad + u + W]
            j = jobs[t][s * NFEAT:(s + 1) * NFEAT]
            tx, ty = pts[:, 0] - u, pts[:, 1] - v
            j["frame"] = s
            j["projw"] = j["projh"] = TSIZE
            j["proj_off"], j["mask_off"] = offs[:, 0], offs[:, 1]
            j["px"] = tx + rng.integers(-3, 4, NFEAT) + rng.random(NFEAT) * 0.9
            j["py"] = ty + rng.integers(-3, 4, NFEAT) + rng.random(NFEAT) * 0.9
            j["threshold"] = THRESHOLD
            j["stride"] = STRIDE
            j["amask_off"] = -1
            truth[t][s * NFEAT:(s + 1) * NFEAT, 0] = tx
            truth[t][s * NFEAT:(s + 1) * NFEAT, 1] = ty
    return frames, jobs, np.concatenate(blob_parts), truth


def oracle_run(frames, jobs, blob, seqs):
    """The CPU oracle on the frames `seqs` (indices) of one time-step; returns (results, seconds, positions)."""
    from oracle import pyoracle as O
    O.build()
    lc = level_cells_and_bytes()
    res = []
    n = TSIZE * TSIZE
    t0 = time.perf_counter()
    for f in seqs:
        levels = O.pyramid(frames[f], SUBS)
        for k in range(NFEAT):
            j = jobs[f * NFEAT + k]
            proj = blob[j["proj_off"]:j["proj_off"] + n].reshape(TSIZE, TSIZE)
            mask = blob[j["mask_off"]:j["mask_off"] + n].reshape(TSIZE, TSIZE)
            res.append(O.track(levels, SUBS, proj, mask, float(j["px"]), float(j["py"]), THRESHOLD, STRIDE))
    dt = time.perf_counter() - t0
    pos = sum(sum(l["cells"] for l in lc[:r["levels_run"]]) for r in res)
    return res, dt, pos


def _oracle_worker(args):
    """One process of the all-cores CPU baseline: tracks a slice of frames with the oracle."""
    frames, jobs, blob, reps = args
    dt = pos = 0
    for _ in range(reps):
        _, d, p = oracle_run(frames, jobs, blob, range(len(frames)))
        dt += d; pos += p
    return dt, pos


def oracle_run_parallel(frames, jobs, blob, nfr, nproc, reps=6):
    """The same oracle pass split over nproc worker processes (spawned: they never touch the GPU), each repeating
    its slice `reps` times.  Returns (wall seconds of the parallel section, positions)."""
    import multiprocessing as mp
    parts = []
    for k in range(nproc):
        lo, hi = nfr * k // nproc, nfr * (k + 1) // nproc
        if hi > lo:
            jj = jobs[lo * NFEAT:hi * NFEAT].copy()
            jj["frame"] -= lo
            parts.append((frames[lo:hi], jj, blob, reps))
    with mp.get_context("spawn").Pool(len(parts)) as pool:
        pool.map(_oracle_worker, [(p[0][:1], p[1][:NFEAT], p[2], 1) for p in parts])      # start-up and page-in, untimed
        t0 = time.perf_counter()
        res = pool.map(_oracle_worker, parts)
        wall = time.perf_counter() - t0
    return wall, sum(r[1] for r in res)


def same_track(g, r):
    """GPU record g equals oracle result r: position, levels walked and the response bit for bit."""
    return (g["x"], g["y"], g["levels_run"]) == (r["x"], r["y"], r["levels_run"]) and \
        np.float32(g["response"]).view(np.uint32) == np.float32(r["response"]).view(np.uint32)


LDS_PROFILE = {}


def load_traffic():
    """HBM bytes per launch of the kernels of this command, from the committed rocprofv3 --pmc passes (FETCH_SIZE and
    WRITE_SIZE in separate runs, read counter x2: profiles/tools/profile_bench.sh).  A profile of another workload
    shape or an older kernel does not apply: the file names the shapes it was taken on.  The same file carries the
    kernels' LDS-array counters (lds_report)."""
    if not os.path.exists(TRAFFIC_FILE):
        return {}
    with open(TRAFFIC_FILE) as fh:
        doc = json.load(fh)
    LDS_PROFILE.clear()
    LDS_PROFILE.update(doc.get("lds", {}))
    return doc.get("kernels", {})


def lds_report(ms, *keys):
    """LDS figures of a leg from the committed counter pass (profiles/<round>_lds.txt): the share of all CU cycles in
    which the LDS array was active, the share of those cycles that were bank-conflict cycles, and an upper bound on the
    bytes per second through the array (conflict-free array cycles x 256 B, over this run's kernel time) against the
    aggregate ds_read_b64 peak.  None when the profile does not hold the kernels."""
    if not all(k in LDS_PROFILE for k in keys):
        return None
    a = sum(LDS_PROFILE[k]["array_cycles"] for k in keys)
    c = sum(LDS_PROFILE[k]["conflict_cycles"] for k in keys)
    cyc = sum(LDS_PROFILE[k]["cycles_per_xcd"] for k in keys)
    bps = (a - c) * 256.0 / (ms * 1e-3)
    return {"array_busy": a / (cyc * 256.0) if cyc else None, "conflict_share": c / a if a else None,
            "bytes_per_s_upper_bound": bps, "peak_bytes_per_s": LDS_PEAK_BPS, "frac_of_peak": bps / LDS_PEAK_BPS,
            "source": os.path.basename(TRAFFIC_FILE)}


def traffic_of(traffic, *keys):
    """Sum of the profiled HBM bytes per launch of the kernels `name@workgroups`; None when one of them is not in the profile."""
    if not all(k in traffic for k in keys):
        return None
    return sum(traffic[k]["hbm_bytes_per_launch"] for k in keys)


class Ms(float):
    """Milliseconds of one launch: the median over the repeats, with the fastest repeat beside it."""
    fastest = None


def timed(stream, fn, steps, warm=5):
    """Median (and, as .fastest, minimum) milliseconds of fn() between HIP events recorded on the stream the kernels
    are launched on, over at least 30 launches."""
    import torch
    for _ in range(warm):
        fn()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(max(steps, 30))]
    for a, b in evs:
        a.record(stream); fn(); b.record(stream)
    torch.cuda.synchronize()
    t = [a.elapsed_time(b) for a, b in evs]
    ms = Ms(np.median(t))
    ms.fastest = float(np.min(t))
    return ms


def search_report(name, kernel, ms, positions, alg_bytes, macs, traffic, extra=None):
    ach = alg_bytes / (ms * 1e-3) / 1e9
    tr = traffic_of(traffic, *name.split("+"))
    out = {"value": positions / (ms * 1e-3), "unit": "positions/s", "ms": float(ms), "ms_min": getattr(ms, "fastest", None),
           "roofline": {"bound": "hbm", "kernel": kernel, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": ach / HBM_PEAK_GBS, "traffic": tr, "algorithmic_bytes_per_launch": int(alg_bytes),
                        "u8_mac_per_s": macs / (ms * 1e-3),
                        "frac_of_packed_dot_peak": macs / (ms * 1e-3) / DOT4_PEAK_MEASURED,
                        "frac_of_nominal_dot_peak": macs / (ms * 1e-3) / DOT4_PEAK_NOMINAL},
           "lds": lds_report(ms, *name.split("+"))}
    if extra:
        out.update(extra)
    return out


# ---------------------------------------------------------------------------------------------------
# extra legs (rank 0 of a one-GPU run): the other configurations BASELINE.json names
# ---------------------------------------------------------------------------------------------------
def fast_leg(api, capi, ctx, stream, dev, args, traffic):
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

    def go(t=thr):
        capi.check(ctx.handle, L.slam_hip_fast_detect_d(ctx.handle, C.c_void_p(d.data_ptr()), w, h, w, w * h, nfr, t, 1,
                                                        C.c_void_p(d_out.data_ptr()), cap, C.c_void_p(d_cnt.data_ptr()),
                                                        C.c_void_p(d_scr.data_ptr()), sb))
    # the same frames at a threshold that keeps a few thousand corners per frame - the regime getSortedKeyPoints steers
    # fastThreshold into (new_point.cpp:113-127); threshold 10 on this texture is dominated by noise corners
    sparse_thr = 40
    sparse = None
    if not args.no_fast_sparse:
        go(sparse_thr)
        torch.cuda.synchronize()
        sparse_cnt = float(d_cnt.cpu().numpy().mean())
        sparse_ms = timed(stream, lambda: go(sparse_thr), max(5, args.steps))
        sparse = {"threshold": sparse_thr, "ms": float(sparse_ms), "ms_min": sparse_ms.fastest, "value": w * h * nfr / (sparse_ms * 1e-3), "unit": "px/s",
                  "corners_per_frame": sparse_cnt}
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
    ms = timed(stream, go, max(5, args.steps))
    alg = w * h * nfr + 12 * int(np.minimum(cnt, cap).sum())
    p_ntx, p_nty = (w + 247) // 248, (h + 29) // 30                # the tile kernel's grid
    tr = traffic_of(traffic, "fast_tile_kernel@%d" % (p_ntx * p_nty * nfr), "fast_scan_kernel@%d" % nfr, "fast_scatter_kernel@%d" % (p_ntx * p_nty * nfr))
    return {"workload": "C4: %d frames %dx%d, FAST-9-16 threshold %d + non-max suppression; keypoint capacity %d per frame "
                        "(the configuration names 65536: this noise-rich texture has about 118 k corners per frame at threshold 10, "
                        "so the larger capacity keeps every list whole; the sparse figure runs the same frames at threshold %d)"
                        % (nfr, w, h, thr, cap, sparse_thr),
            "value": w * h * nfr / (ms * 1e-3), "unit": "px/s", "ms": float(ms), "ms_min": ms.fastest, "corners_per_frame": float(cnt.mean()),
            "roofline": {"bound": "hbm", "kernel": "fast_tile_kernel+fast_scan_kernel+fast_scatter_kernel", "achieved": alg / (ms * 1e-3) / 1e9,
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": alg / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                         "traffic": tr, "algorithmic_bytes_per_launch": alg},
            "sparse": sparse,
            "parity": parity}


def tile_side(a, p, size, x0):
    """Image pixels under all candidate patches along one axis: [max(x0,0), min(size, x0+a+p-1))."""
    return min(size, x0 + a + p - 1) - max(x0, 0)


def c1_leg(api, capi, ctx, stream, dev, args, traffic):
    """C1 (BASELINE.json configs[0]): 320x240, 16 features per frame, 11x11 templates, single level, 41x41 windows
    (fullResSearchRad 20), stride 1, the four mask modes of matchFeatureInArea mixed; 1024 frames."""
    import torch
    from slam_amd import synth
    w, h, nfr, nfeat, ts, win = 320, 240, 1024, 16, 11, 41
    rng = np.random.default_rng(0x51A40001)
    f1 = synth.frame(w, h, synth.SEED_BASE + 1)
    pts = synth.feature_grid(w, h, nfeat, 30, rng)
    parts, off, per_feat = [], 0, []
    for k, (cx, cy) in enumerate(pts):
        patch = synth.cut_template(f1, cx, cy, ts, ts)
        pm = rng.integers(0, 2, (ts, ts), dtype=np.uint8)
        am = synth.ellipse_mask(win, win, 18.0, 9.0, 0.3 * k)
        per_feat.append((off, off + patch.size, off + 2 * patch.size, patch, pm, am))
        parts += [patch.reshape(-1), pm.reshape(-1), am.reshape(-1)]
        off += 2 * patch.size + am.size
    blob = np.concatenate(parts)
    jobs = np.zeros(nfr * nfeat, MATCH_JOB_DT)
    f_idx, k_idx = np.divmod(np.arange(nfr * nfeat), nfeat)
    mode = (f_idx + k_idx) & 3
    po = np.array([p[0] for p in per_feat]); mo = np.array([p[1] for p in per_feat]); ao = np.array([p[2] for p in per_feat])
    cxs = np.array([p[0] for p in pts]); cys = np.array([p[1] for p in pts])
    jobs["frame"], jobs["level"], jobs["pw"], jobs["ph"] = f_idx, 0, ts, ts
    jobs["patch_off"] = po[k_idx]
    jobs["pmask_off"] = np.where(mode & 1, mo[k_idx], -1)
    jobs["ax"], jobs["ay"], jobs["aw"], jobs["ah"] = cxs[k_idx] - 20, cys[k_idx] - 20, win, win
    jobs["amask_off"] = np.where(mode & 2, ao[k_idx], -1)
    jobs["stride"], jobs["use_exact_area_mask"], jobs["score_off"] = 1, 0, -1
    d_fr = torch.from_numpy(f1).to(dev).unsqueeze(0).repeat(nfr, 1, 1).contiguous()
    pyr = api.ImagePyramid(ctx, None, [1.0], device_ptr=d_fr.data_ptr(), pitch=w, frame_stride=w * h, shape=(nfr, h, w))
    d_blob = torch.from_numpy(blob).to(dev)
    d_jobs = torch.from_numpy(jobs.view(np.uint8).reshape(-1)).to(dev)
    d_res = torch.zeros(len(jobs) * MATCH_RES_DT.itemsize, dtype=torch.uint8, device=dev)
    L = capi.lib()
    P = lambda t: C.c_void_p(t.data_ptr())

    def go():
        capi.check(ctx.handle, L.slam_hip_match_batch_d(ctx.handle, pyr.handle, P(d_jobs), len(jobs), P(d_blob), blob.size,
                                                        None, 0, P(d_res)))
    go()
    torch.cuda.synchronize()
    res = d_res.cpu().numpy().view(MATCH_RES_DT)
    assert (res["status"] == 0).all()
    # cells, bytes and multiply-accumulates per (feature, mode): from the geometry, and the oracle's score map when present
    cells = np.zeros((nfeat, 4)); byts = np.zeros((nfeat, 4)); parity = None
    for k, (cx, cy) in enumerate(pts):
        tw_, th_ = tile_side(win, ts, w, cx - 20 - ts // 2), tile_side(win, ts, h, cy - 20 - ts // 2)
        for m in range(4):
            byts[k, m] = tw_ * th_ + ts * ts * (1 + (m & 1)) + win * win * ((m >> 1) & 1) + 20
            am = per_feat[k][5]
            # stride 1, window inside the image: every cell, except those the area mask removes in the patch-mask modes
            cells[k, m] = int(am.sum()) if (m & 2) and (m & 1) else win * win
    if not args.no_cpu:
        from oracle import pyoracle as O
        O.build()
        for k in range(nfeat):
            _, _, _, patch, pm, am = per_feat[k]
            for m in range(4):
                r = O.match(f1, patch, pm if m & 1 else None, (pts[k][0] - 20, pts[k][1] - 20, win, win), am if m & 2 else None, 1,
                            False, want_score=True)
                assert int((r["score"] != -1).sum()) == cells[k, m], (k, m)
                f = (m - k) & 3                                   # a frame on which feature k runs in mode m
                g = res[f * nfeat + k]
                if (g["x"], g["y"]) != (r["x"], r["y"]) or np.float32(g["response"]).view(np.uint32) != np.float32(r["response"]).view(np.uint32):
                    raise SystemExit("PARITY FAILURE (C1) feature %d mode %d: gpu %s oracle %s" % (k, m, g, r))
        same = res.reshape(nfr, nfeat)
        for f in range(4, nfr):                                    # all frames are the same image: results repeat with period 4
            assert np.array_equal(same[f], same[f & 3])
        parity = "every (feature, mask mode) pair equals the oracle bit for bit; all %d jobs repeat those" % len(jobs)
    deferred = C.c_int32(0)
    capi.check(ctx.handle, L.slam_hip_debug_walk_deferred(ctx.handle, C.byref(deferred)))
    ms = timed(stream, go, max(5, args.steps))
    positions = float(cells[k_idx, mode].sum())
    alg = float(byts[k_idx, mode].sum())
    out = search_report("match_lean_kernel@%d+match_kernel@%d" % (len(jobs), min(len(jobs), 4096)), "match_lean_kernel+match_kernel", ms, positions, alg, positions * 2 * ts * ts, traffic,
                        {"workload": "C1: %dx%d, %d features/frame, %dx%d templates, %dx%d windows, stride 1, four mask modes mixed; %d frames"
                                     % (w, h, nfeat, ts, ts, win, win, nfr), "jobs": len(jobs),
                         "jobs_left_to_general_kernel": deferred.value, "parity": parity})
    pyr.close()
    return out


def c3_leg(api, capi, ctx, stream, dev, args, traffic):
    """C3 (BASELINE.json configs[2]): 1280x720, 256 features per frame, 31x31 masked templates, pyramid {1,2,4,8};
    the coarsest level is searched under a rasterised uncertainty ellipse (patch mask + area mask mode,
    matching.cpp:60-104), finer levels by Feature::track's window rule; 256 frames."""
    import torch
    from slam_amd import synth
    w, h, nfr, nfeat, ts, subs = 1280, 720, 256, 256, 31, [1.0, 2.0, 4.0, 8.0]
    rng = np.random.default_rng(0x51A40003)
    shift = (5, -4)
    f0, f1 = synth.frame(w, h, synth.SEED_BASE + 33), synth.frame(w, h, synth.SEED_BASE + 33, shift)
    pts = synth.feature_grid(w, h, nfeat, 130, rng)          # every window of the walk stays inside the image
    lc = level_cells_and_bytes(ts, subs, STRIDE)             # levels 3 (replaced below), 2, 1, 0
    parts, off, feats = [], 0, []
    for (cx, cy) in pts:
        proj = synth.cut_template(f0, cx, cy, ts, ts)
        mask = np.ones_like(proj); mask[0, :5] = 0; mask[-1, -5:] = 0
        a, b, th = float(rng.uniform(24, 96)) / 8, float(rng.uniform(12, 48)) / 8, float(rng.uniform(0, np.pi))
        ew = 2 * int(np.ceil(max(a, b))) + 1
        am = synth.ellipse_mask(ew, ew, a, b, th)
        px, py = cx - shift[0] + float(rng.integers(-10, 11)) + 0.4, cy - shift[1] + float(rng.integers(-6, 7)) - 0.3
        ax, ay = int(round(px / 8)) - ew // 2, int(round(py / 8)) - ew // 2
        tw3 = lc[0]["tw"]
        cells3 = int(am[0::STRIDE, 0::STRIDE].sum())         # interior lattice from (0, 0), cells the area mask keeps
        bytes3 = (ew + tw3 - 1) ** 2 + 2 * tw3 * tw3 + ew * ew + 20
        feats.append(dict(po=off, mo=off + proj.size, ao=off + 2 * proj.size, ew=ew, px=px, py=py, ax=ax, ay=ay,
                          proj=proj, mask=mask, am=am, cells3=cells3, bytes3=bytes3))
        parts += [proj.reshape(-1), mask.reshape(-1), am.reshape(-1)]
        off += 2 * proj.size + am.size
    blob = np.concatenate(parts)
    jobs = np.zeros(nfr * nfeat, TRACK_JOB_DT)
    f_idx, k_idx = np.divmod(np.arange(nfr * nfeat), nfeat)
    col = lambda key: np.array([ft[key] for ft in feats])[k_idx]
    jobs["frame"], jobs["projw"], jobs["projh"], jobs["flags"] = f_idx, ts, ts, capi.TRACK_COARSE_AREA
    jobs["proj_off"], jobs["mask_off"], jobs["amask_off"] = col("po"), col("mo"), col("ao")
    jobs["px"], jobs["py"], jobs["threshold"], jobs["stride"] = col("px"), col("py"), 0.9, STRIDE
    jobs["ax"], jobs["ay"], jobs["aw"], jobs["ah"] = col("ax"), col("ay"), col("ew"), col("ew")
    d_fr = torch.from_numpy(f1).to(dev).unsqueeze(0).repeat(nfr, 1, 1).contiguous()
    pyr = api.ImagePyramid(ctx, None, subs, device_ptr=d_fr.data_ptr(), pitch=w, frame_stride=w * h, shape=(nfr, h, w))
    d_blob = torch.from_numpy(blob).to(dev)
    d_jobs = torch.from_numpy(jobs.view(np.uint8).reshape(-1)).to(dev)
    d_res = torch.zeros(len(jobs) * TRACK_RES_DT.itemsize, dtype=torch.uint8, device=dev)
    L = capi.lib()
    P = lambda t: C.c_void_p(t.data_ptr())

    def go():
        capi.check(ctx.handle, L.slam_hip_track_batch_d(ctx.handle, pyr.handle, P(d_jobs), len(jobs), P(d_blob), blob.size, P(d_res)))
    # 31x31 templates: the lean walk kernel needs 11 KB of LDS per walk for the full-resolution level (include/slam_hip.h)
    capi.check(ctx.handle, L.slam_hip_set_walk_lds(ctx.handle, args.c3_walk_lds))     # 10240: 8 LDS granules of 1280 bytes, 16 walks per CU
    capi.check(ctx.handle, L.slam_hip_set_walk_chunks(ctx.handle, int(args.c3_chunks)))
    go()
    torch.cuda.synchronize()
    res = d_res.cpu().numpy().view(TRACK_RES_DT)
    deferred = C.c_int32(0)
    capi.check(ctx.handle, L.slam_hip_debug_walk_deferred(ctx.handle, C.byref(deferred)))
    assert (res["status"] == 0).all()
    parity = None
    if not args.no_cpu:
        from oracle import pyoracle as O
        O.build()
        levels = O.pyramid(f1, subs)
        for k, ft in enumerate(feats):
            r = O.track(levels, subs, ft["proj"], ft["mask"], ft["px"], ft["py"], 0.9, STRIDE,
                        coarse_area=(ft["ax"], ft["ay"], ft["ew"], ft["ew"]), coarse_mask=ft["am"])
            if not same_track(res[k], r):
                raise SystemExit("PARITY FAILURE (C3) feature %d: gpu %s oracle %s" % (k, res[k], r))
        same = res.reshape(nfr, nfeat)
        for f in range(1, nfr):
            assert np.array_equal(same[f], same[0])
        parity = "all %d features of a frame equal the oracle bit for bit; the %d frames repeat them" % (nfeat, nfr)
    # positions / bytes / MACs by the levels each walk ran
    lr = res["levels_run"]
    c3 = col("cells3").astype(np.float64); b3 = col("bytes3").astype(np.float64)
    tw3 = lc[0]["tw"]
    cum_c = np.array([0] + list(np.cumsum([l["cells"] for l in lc[1:]])), np.float64)
    cum_b = np.array([0] + list(np.cumsum([l["bytes"] for l in lc[1:]])), np.float64)
    cum_m = np.array([0] + list(np.cumsum([l["macs"] for l in lc[1:]])), np.float64)
    ran = lr >= 1
    positions = float((ran * c3 + cum_c[np.maximum(lr - 1, 0)]).sum())
    alg = float((ran * b3 + cum_b[np.maximum(lr - 1, 0)]).sum())
    macs = float((ran * c3 * 2 * tw3 * tw3 + cum_m[np.maximum(lr - 1, 0)]).sum())
    ms = timed(stream, go, max(5, args.steps))
    ms_p = timed(stream, lambda: pyr.rebuild_d(d_fr.data_ptr(), w, w * h), max(5, args.steps))
    out = search_report("track_lean_kernel<%s>@%d+track_kernel@4096" % ("true" if args.c3_chunks else "false", len(jobs)),
                        "track_lean_kernel+track_kernel", ms, positions, alg, macs, traffic,
                        {"workload": "C3: %dx%d, %d features/frame, %dx%d masked templates, pyramid {1,2,4,8}, ellipse windows "
                                     "(area masks) on the coarsest level, stride 3; %d frames" % (w, h, nfeat, ts, ts, nfr),
                         "jobs": len(jobs), "pyramid_ms": ms_p, "frames_per_s_with_pyramid": nfr / ((ms + ms_p) * 1e-3),
                         "walked_all_levels": float((lr == 4).mean()), "jobs_left_to_general_kernel": deferred.value,
                         "walk_lds_bytes": args.c3_walk_lds, "walk_chunks": bool(args.c3_chunks), "parity": parity})
    capi.check(ctx.handle, L.slam_hip_set_walk_chunks(ctx.handle, 0))
    capi.check(ctx.handle, L.slam_hip_set_walk_lds(ctx.handle, 7680))      # back to the defaults
    pyr.close()
    return out


def prod_leg(api, capi, ctx, stream, dev, args, traffic):
    """The reference's live point-tracking shape (not a BASELINE configuration): matchPoints' pyramid {1, 3} and stride 3
    (matching.cpp:243-247) with the 81x81 descriptors new_image.cpp:80-82 creates, 64 features per 640x480 frame, masks
    with cut corners (a warped quadrilateral), Feature::track's windows (feature.cpp:133-146): 14x14 on the stride-3
    lattice at sub = 3 (27x27 template), 24x24 at full resolution; 256 frames."""
    import torch
    from slam_amd import synth
    w, h, nfr, nfeat, ts, subs = W, H, 256, 64, 81, [1.0, 3.0]
    rng = np.random.default_rng(0x51A40006)
    shift = (4, -3)
    big = synth.canvas_smooth(-8, -8, w + 16, h + 16, 0x51A40006)
    f0, f1 = big[8:8 + h, 8:8 + w], big[8 + shift[1]:8 + shift[1] + h, 8 + shift[0]:8 + shift[0] + w]
    f0, f1 = np.ascontiguousarray(f0), np.ascontiguousarray(f1)
    pts = synth.feature_grid(w, h, nfeat, 70, rng)           # every window of the walk stays inside the image
    lc = level_cells_and_bytes(ts, subs, STRIDE)             # levels 1, 0
    parts, off, feats = [], 0, []
    for (cx, cy) in pts:
        proj = synth.cut_template(f0, cx, cy, ts, ts)
        mask = np.ones_like(proj)
        c = int(rng.integers(0, ts // 3 + 1))
        for r in range(c):
            mask[r, :c - r] = 0
            mask[ts - 1 - r, ts - (c - r):] = 0
        px, py = cx - shift[0] + float(rng.integers(-3, 4)) + 0.4, cy - shift[1] + float(rng.integers(-3, 4)) - 0.3
        feats.append(dict(po=off, mo=off + proj.size, px=px, py=py, proj=proj, mask=mask))
        parts += [proj.reshape(-1), mask.reshape(-1)]
        off += 2 * proj.size
    blob = np.concatenate(parts)
    jobs = np.zeros(nfr * nfeat, TRACK_JOB_DT)
    f_idx, k_idx = np.divmod(np.arange(nfr * nfeat), nfeat)
    col = lambda key: np.array([ft[key] for ft in feats])[k_idx]
    jobs["frame"], jobs["projw"], jobs["projh"] = f_idx, ts, ts
    jobs["proj_off"], jobs["mask_off"], jobs["amask_off"] = col("po"), col("mo"), -1
    jobs["px"], jobs["py"], jobs["threshold"], jobs["stride"] = col("px"), col("py"), THRESHOLD, STRIDE
    d_fr = torch.from_numpy(f1).to(dev).unsqueeze(0).repeat(nfr, 1, 1).contiguous()
    pyr = api.ImagePyramid(ctx, None, subs, device_ptr=d_fr.data_ptr(), pitch=w, frame_stride=w * h, shape=(nfr, h, w))
    d_blob = torch.from_numpy(blob).to(dev)
    d_jobs = torch.from_numpy(jobs.view(np.uint8).reshape(-1)).to(dev)
    d_res = torch.zeros(len(jobs) * TRACK_RES_DT.itemsize, dtype=torch.uint8, device=dev)
    L = capi.lib()
    P = lambda t: C.c_void_p(t.data_ptr())

    # 81x81 descriptors do not fit the lean walk kernel's LDS whole: its build that takes the template in chunks of rows,
    # with 10 granules of 1280 bytes per walk (12 walks per CU); 0 leaves every walk to the general kernel
    if args.prod_walk_lds:
        capi.check(ctx.handle, L.slam_hip_set_walk_lds(ctx.handle, args.prod_walk_lds))
        capi.check(ctx.handle, L.slam_hip_set_walk_chunks(ctx.handle, 1))

    def go():
        capi.check(ctx.handle, L.slam_hip_track_batch_d(ctx.handle, pyr.handle, P(d_jobs), len(jobs), P(d_blob), blob.size, P(d_res)))
    go()
    torch.cuda.synchronize()
    res = d_res.cpu().numpy().view(TRACK_RES_DT)
    deferred = C.c_int32(0)
    capi.check(ctx.handle, L.slam_hip_debug_walk_deferred(ctx.handle, C.byref(deferred)))
    assert (res["status"] == 0).all()
    parity = None
    if not args.no_cpu:
        from oracle import pyoracle as O
        O.build()
        levels = O.pyramid(f1, subs)
        for k, ft in enumerate(feats):
            r = O.track(levels, subs, ft["proj"], ft["mask"], ft["px"], ft["py"], THRESHOLD, STRIDE)
            if not same_track(res[k], r):
                raise SystemExit("PARITY FAILURE (prod) feature %d: gpu %s oracle %s" % (k, res[k], r))
        same = res.reshape(nfr, nfeat)
        for f in range(1, nfr):
            assert np.array_equal(same[f], same[0])
        parity = "all %d features of a frame equal the oracle bit for bit; the %d frames repeat them" % (nfeat, nfr)
    lr = res["levels_run"]
    cum_c = np.array([0] + list(np.cumsum([l["cells"] for l in lc])), np.float64)
    cum_b = np.array([0] + list(np.cumsum([l["bytes"] for l in lc])), np.float64)
    cum_m = np.array([0] + list(np.cumsum([l["macs"] for l in lc])), np.float64)
    positions, alg, macs = float(cum_c[lr].sum()), float(cum_b[lr].sum()), float(cum_m[lr].sum())
    tx = np.array([p[0] for p in pts])[k_idx] - shift[0]
    ty = np.array([p[1] for p in pts])[k_idx] - shift[1]
    tracked = float(np.mean((res["x"] == tx) & (res["y"] == ty) & (res["response"] > THRESHOLD)))
    ms = timed(stream, go, max(5, args.steps))
    ms_p = timed(stream, lambda: pyr.rebuild_d(d_fr.data_ptr(), w, w * h), max(5, args.steps))

    def one_frame():                                             # the reference's own call: the features of one frame
        capi.check(ctx.handle, L.slam_hip_track_batch_d(ctx.handle, pyr.handle, P(d_jobs), nfeat, P(d_blob), blob.size, P(d_res)))
    lat = timed(stream, one_frame, max(5, args.steps))
    go()
    out = search_report("track_lean_kernel<%s>@%d+track_kernel@4096" % ("true" if args.prod_walk_lds else "false", len(jobs)),
                        "track_lean_kernel+track_kernel", ms, positions, alg, macs, traffic,
                        {"workload": "the reference's live shape: %dx%d, %d features/frame, %dx%d masked descriptors, pyramid {1,3}, "
                                     "stride 3, threshold %.2f; %d frames" % (w, h, nfeat, ts, ts, THRESHOLD, nfr),
                         "jobs": len(jobs), "pyramid_ms": float(ms_p), "frames_per_s_with_pyramid": nfr / ((ms + ms_p) * 1e-3),
                         "walked_all_levels": float((lr == 2).mean()), "tracked_fraction": tracked,
                         "jobs_left_to_general_kernel": deferred.value, "parity": parity,
                         "walk_lds_bytes": args.prod_walk_lds or 7680, "walk_chunks": bool(args.prod_walk_lds),
                         "one_frame": {"features": nfeat, "ms": float(lat), "ms_min": lat.fastest,
                                       "note": "launch to completion of one frame's walks between HIP events (with the chunked build: "
                                               "the last level of every walk in 8 bands of rows, one wavefront each)"}})
    capi.check(ctx.handle, L.slam_hip_set_walk_chunks(ctx.handle, 0))
    capi.check(ctx.handle, L.slam_hip_set_walk_lds(ctx.handle, 7680))      # back to the defaults
    pyr.close()
    return out


def line_leg(api, capi, ctx, stream, dev, args, traffic):
    """The searches of LineFeature::track (lineFeature.cpp:120-165) as matchLines runs them (matching.cpp:273-277:
    pyramid {1,2}, stride 2): 3 lines per frame (new_image.cpp:80), 81x81 descriptors (new_image.cpp:82), the coarsest
    level searched without a patch mask inside a 60x40 disc-union area mask, then a 5x5 window at full resolution."""
    import torch
    from slam_amd import synth
    w, h, nfr, nline, half, subs = 640, 480, 1024, 3, 40, [1.0, 2.0]
    ds = 2 * half + 1
    aw, ah = 60, 40
    f0 = synth.frame(w, h, 5)
    f1 = synth.frame(w, h, 5, (3, -2))
    centres = [(160, 140), (330, 250), (500, 330)]
    descs = [synth.cut_template(f0, cx, cy, ds, ds) for (cx, cy) in centres]
    yy, xx = np.mgrid[0:ah, 0:aw]
    masks, areas = [], []
    for k, (cx, cy) in enumerate(centres):
        m = np.zeros((ah, aw), np.uint8)
        for b in range(8):                                   # discs along the projected cone (cv::circle filled: d^2 <= r^2)
            ccx, ccy, r = 6 + 6 * b, 8 + 3 * b, 4 + (b + k) % 4
            m[(xx - ccx) ** 2 + (yy - ccy) ** 2 <= r * r] = 255      # lineFeature.cpp:95 draws 255
        masks.append(m)
        areas.append(((cx - 3) // 2 - 27, (cy + 2) // 2 - 17))
    doff = [k * ds * ds for k in range(nline)]
    moff = [nline * ds * ds + k * aw * ah for k in range(nline)]
    blob = np.concatenate([d.reshape(-1) for d in descs] + [m.reshape(-1) for m in masks])
    jobs = np.zeros(nfr * nline, LINE_JOB_DT)
    f_idx, k_idx = np.divmod(np.arange(nfr * nline), nline)
    jobs["frame"], jobs["dw"], jobs["dh"], jobs["stride"] = f_idx, ds, ds, 2
    jobs["desc_off"], jobs["amask_off"] = np.array(doff)[k_idx], np.array(moff)[k_idx]
    jobs["ax"], jobs["ay"] = np.array([a[0] for a in areas])[k_idx], np.array([a[1] for a in areas])[k_idx]
    jobs["aw"], jobs["ah"], jobs["threshold"] = aw, ah, 0.9
    d_fr = torch.from_numpy(f1).to(dev).unsqueeze(0).repeat(nfr, 1, 1).contiguous()
    pyr = api.ImagePyramid(ctx, None, subs, device_ptr=d_fr.data_ptr(), pitch=w, frame_stride=w * h, shape=(nfr, h, w))
    d_blob = torch.from_numpy(blob).to(dev)
    d_jobs = torch.from_numpy(jobs.view(np.uint8).reshape(-1)).to(dev)
    d_res = torch.zeros(len(jobs) * TRACK_RES_DT.itemsize, dtype=torch.uint8, device=dev)
    L = capi.lib()
    P = lambda t: C.c_void_p(t.data_ptr())

    def go():
        capi.check(ctx.handle, L.slam_hip_line_track_batch_d(ctx.handle, pyr.handle, P(d_jobs), len(jobs), P(d_blob), blob.size, P(d_res)))
    go()
    torch.cuda.synchronize()
    res = d_res.cpu().numpy().view(TRACK_RES_DT)
    assert (res["status"] == 0).all()
    parity = None
    if not args.no_cpu:
        from oracle import pyoracle as O
        O.build()
        levels = O.pyramid(f1, subs)
        for k in range(nline):
            r = O.line_track(levels, subs, descs[k], (areas[k][0], areas[k][1], aw, ah), masks[k], 0.9, 2)
            if not same_track(res[k], r):
                raise SystemExit("PARITY FAILURE (line walk) line %d: gpu %s oracle %s" % (k, res[k], r))
        same = res.reshape(nfr, nline)
        for f in range(1, nfr):
            assert np.array_equal(same[f], same[0])
        parity = "the %d lines of a frame equal the oracle bit for bit; the %d frames repeat them" % (nline, nfr)
    # coarsest level: no patch mask -> every interior cell through matchTemplate (matching.cpp:151-153), 40x40 template;
    # full resolution: searchRad 2 -> 5x5 cells, 81x81 template
    tc = ds // 2
    lr = res["levels_run"]
    c = np.array([0, aw * ah, aw * ah + 25], np.float64)
    b = np.array([0, (aw + tc - 1) * (ah + tc - 1) + tc * tc + aw * ah + 20,
                  (aw + tc - 1) * (ah + tc - 1) + tc * tc + aw * ah + 20 + (5 + ds - 1) ** 2 + ds * ds + 20], np.float64)
    m = np.array([0, 2.0 * aw * ah * tc * tc, 2.0 * aw * ah * tc * tc + 2.0 * 25 * ds * ds])
    ms = timed(stream, go, max(5, args.steps))
    # one frame's lines alone (the reference's own call pattern: three lines per frame): the coarsest search of each is
    # cut into bands of rows over several wavefronts; beside it the same call with one wavefront per line
    def one_frame(c):
        def f():
            capi.check(c.handle, L.slam_hip_line_track_batch_d(c.handle, pyr.handle, P(d_jobs), nline, P(d_blob), blob.size, P(d_res)))
        return f
    lat = timed(stream, one_frame(ctx), max(5, args.steps))
    keep = os.environ.get("SLAM_HIP_LINE_BANDS")
    os.environ["SLAM_HIP_LINE_BANDS"] = "1"
    ctx1 = api.Context(ctx.device, stream=stream.cuda_stream)
    if keep is None:
        del os.environ["SLAM_HIP_LINE_BANDS"]
    else:
        os.environ["SLAM_HIP_LINE_BANDS"] = keep
    lat1 = timed(stream, one_frame(ctx1), max(5, args.steps))
    ctx1.close()
    go()                                                          # d_res holds the whole batch again
    out = search_report("line_lean_coarse_kernel@%d+line_track_kernel@%d" % (len(jobs), len(jobs)), "line_lean_coarse_kernel+line_track_kernel", ms, float(c[lr].sum()), float(b[lr].sum()), float(m[lr].sum()), traffic,
                        {"workload": "line walk: %dx%d, %d lines/frame, %dx%d descriptors, pyramid {1,2}, stride 2, %dx%d disc-union "
                                     "area on the coarsest level; %d frames" % (w, h, nline, ds, ds, aw, ah, nfr),
                         "jobs": len(jobs), "refined_at_full_resolution": float((lr == 2).mean()), "parity": parity,
                         "one_frame": {"lines": nline, "ms": float(lat), "ms_min": lat.fastest,
                                       "ms_one_wavefront_per_line": float(lat1), "bands_per_line": min(8, max(1, 3072 // nline)),
                                       "note": "launch to completion of one frame's lines between HIP events"}})
    pyr.close()
    return out


# ---------------------------------------------------------------------------------------------------
def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    args = parse_args(argv)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(launch_ranks(args, argv))

    import torch
    import torch.distributed as dist
    from slam_amd import api, capi, parallel

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
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

    nseq, nts = args.sequences_per_gpu, args.time_steps
    lo, hi = parallel.shard_range(nseq * world, rank, world)          # this rank's sequences of the whole job
    rng = np.random.default_rng(0x51A40002 + 7919 * rank)
    frames, jobs, blob, truth = build_time_steps(hi - lo, nts, 0x51A40002 + lo, rng)
    njobs = nseq * NFEAT

    # inputs resident in HBM before the timed region
    d_frames = torch.from_numpy(frames).to(dev)
    d_jobs = [torch.from_numpy(j.view(np.uint8).reshape(-1)).to(dev) for j in jobs]
    d_blob = torch.from_numpy(blob).to(dev)
    d_out = [torch.zeros(njobs * TRACK_RES_DT.itemsize, dtype=torch.uint8, device=dev) for _ in range(nts)]
    stream = torch.cuda.Stream(device=dev)      # a real (non-null) stream shared by the walk kernels and the events
    torch.cuda.set_stream(stream)
    ctx = api.Context(local_rank, stream=stream.cuda_stream)
    fbytes = nseq * W * H
    L = capi.lib()
    h_frames = torch.from_numpy(frames).pin_memory() if args.with_h2d else None
    # The pyramid build is HBM-bound and the walk VALU-bound: with the pyramid of time-step k+1 built on a second stream
    # while time-step k is walked they share the chip.  Two pyramids, written and read in turn; events order the streams.
    overlap = not args.no_overlap and not args.with_h2d
    mk_pyr = lambda c: api.ImagePyramid(c, None, SUBS, device_ptr=d_frames.data_ptr(), pitch=W, frame_stride=W * H, shape=(nseq, H, W))
    if overlap:
        stream_b = torch.cuda.Stream(device=dev)
        ctx_b = api.Context(local_rank, stream=stream_b.cuda_stream)
        pyrs = [mk_pyr(ctx_b), mk_pyr(ctx_b)]
        built = [torch.cuda.Event(), torch.cuda.Event()]
        walked = [torch.cuda.Event(), torch.cuda.Event()]
    else:
        stream_b, ctx_b = stream, ctx
        pyrs = [mk_pyr(ctx)]
    pyr = pyrs[0]
    issued = {"for": -1}                        # the step whose pyramid has been put on the second stream already

    def build(k):
        p, t = k & 1, k % nts
        stream_b.wait_event(walked[p])          # the walk that last read this pyramid (no-op before the first one)
        pyrs[p].rebuild_d(d_frames.data_ptr() + t * fbytes, W, W * H)
        built[p].record(stream_b)
        issued["for"] = k

    def walk(k, p):
        t = k % nts
        capi.check(ctx.handle, L.slam_hip_track_batch_d(ctx.handle, pyrs[p].handle, C.c_void_p(d_jobs[t].data_ptr()), njobs,
                                                        C.c_void_p(d_blob.data_ptr()), blob.size,
                                                        C.c_void_p(d_out[t].data_ptr())))

    def step(k, ev=None):
        """One time-step: pyramid of frame k % nts of every sequence + the walk of all their features.  ev: a pair of
        timing events put around the walk kernels on their stream."""
        if overlap:
            if issued["for"] != k:
                build(k)
            stream.wait_event(built[k & 1])
            if ev is not None:
                ev[0].record(stream)
            walk(k, k & 1)
            if ev is not None:
                ev[1].record(stream)
            walked[k & 1].record(stream)
            build(k + 1)
            return
        t = k % nts
        if h_frames is not None:
            d_frames[t].copy_(h_frames[t], non_blocking=True)
        pyr.rebuild_d(d_frames.data_ptr() + t * fbytes, W, W * H)
        if ev is not None:
            ev[0].record(stream)
        walk(k, 0)
        if ev is not None:
            ev[1].record(stream)

    # ---- parity gate: results must equal the oracle's before any timing counts ------------------
    for t in range(nts):
        step(t)
    torch.cuda.synchronize()
    res = [d.cpu().numpy().view(TRACK_RES_DT) for d in d_out]
    deferred = C.c_int32(0)
    capi.check(ctx.handle, L.slam_hip_debug_walk_deferred(ctx.handle, C.byref(deferred)))
    for r in res:
        assert (r["status"] == 0).all(), "device rejected jobs"
    lc = level_cells_and_bytes()
    cells = np.array([0] + list(np.cumsum([l["cells"] for l in lc])))
    lbytes = np.array([0] + list(np.cumsum([l["bytes"] for l in lc])))
    lmacs = np.array([0] + list(np.cumsum([l["macs"] for l in lc])))
    positions = [int(cells[r["levels_run"]].sum()) for r in res]            # per time-step
    alg_bytes = float(np.mean([lbytes[r["levels_run"]].sum() for r in res]))
    macs = float(np.mean([lmacs[r["levels_run"]].sum() for r in res]))
    tracked = float(np.mean([np.mean((r["x"] == tr[:, 0]) & (r["y"] == tr[:, 1]) & (r["response"] > THRESHOLD)) for r, tr in zip(res, truth)]))

    cpu = None
    if not args.no_cpu:
        # every rank checks its own results: rank 0 of a one-GPU run the whole first time-step (it is the CPU
        # baseline's sample anyway), every other case a sample of every time-step
        full = world == 1
        cpu_s = cpu_pos = 0.0
        for t in range(nts):
            seqs = range(nseq) if (full and t == 0) else range(t % 7, nseq, max(nseq // 16, 1))
            ores, dt, pos = oracle_run(frames[t], jobs[t], blob, seqs)
            for i, f in enumerate(seqs):
                for k in range(NFEAT):
                    if not same_track(res[t][f * NFEAT + k], ores[i * NFEAT + k]):
                        raise SystemExit("PARITY FAILURE rank %d time-step %d job %d: gpu %s oracle %s"
                                         % (rank, t, f * NFEAT + k, res[t][f * NFEAT + k], ores[i * NFEAT + k]))
            if full and t == 0:
                cpu_s, cpu_pos = dt, pos
        if full:
            passes = 1
            while cpu_s < args.cpu_seconds and passes < 32:       # same sample again until the CPU time is long enough
                _, dt, pos = oracle_run(frames[0], jobs[0], blob, range(nseq))
                cpu_s += dt; cpu_pos += pos; passes += 1
            cpu = {"value": cpu_pos / cpu_s, "unit": "positions/s", "cores": 1, "kind": "port",
                   "sample": "CPU oracle (pyramid + Feature::track) on the %d frames x %d features of the first time-step, "
                             "%d passes, single thread, %.1f s in all; results bit-identical to the GPU's"
                             % (nseq, NFEAT, passes, cpu_s)}
            visible = len(os.sched_getaffinity(0))
            share = int(os.environ.get("OMP_NUM_THREADS", "0") or 0)
            nproc = args.cpu_procs if args.cpu_procs >= 0 else (min(visible, share) if share > 0 else visible)
            if nproc > 1:              # secondary figure: the same port on the host cores this job may use
                wall, pos = oracle_run_parallel(frames[0], jobs[0], blob, nseq, nproc)
                cpu["all_cores"] = {"value": pos / wall, "unit": "positions/s", "cores": nproc,
                                    "host_cores_visible": visible, "host_cpu_share": share if share > 0 else None,
                                    "sample": "same frames split over %d processes, 6 passes each, %.1f s wall" % (nproc, wall)}

    # ---- timing --------------------------------------------------------------------------------
    # A block = exactly `steps` steps between barrier + synchronize on both sides.  One block of 20 steps is 6 ms of
    # GPU time right after seconds of CPU work (the oracle), too short for the chip to reach and hold its clocks, so
    # the block is run `repeats` times back to back (at least 0.25 s of GPU time in all, at least 5 blocks) and the
    # line reports the median block (and the fastest one beside it); every rank runs the same number of blocks.
    def block(first, evs):
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(args.steps):
            step(first + k, evs.get(k))
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        return time.perf_counter() - t0

    for k in range(args.warmup):
        step(k)
    nxt = args.warmup
    cal = block(nxt, {})                          # untimed: calibrates the repeat count, and warms the clocks further
    nxt += args.steps
    repeats = args.repeats if args.repeats > 0 else int(min(max(np.ceil(0.25 / max(cal, 1e-6)), 5), 400))
    repeats = int(parallel.reduce_max_vector([repeats], rdev)[0]) if world > 1 else repeats
    block_s, block_pos, kern = [], [], []
    for b in range(repeats):
        # timing events around the walk kernels of every eighth step: an event pair costs the stream about as much as a
        # small kernel, and the kernels' duration does not need every launch
        evs = {k: tuple(torch.cuda.Event(enable_timing=True) for _ in range(2)) for k in range(0, args.steps, 8)}
        block_s.append(block(nxt, evs))
        block_pos.append(float(sum(positions[(nxt + k) % nts] for k in range(args.steps))))
        kern += [e[0].elapsed_time(e[1]) for e in evs.values()]
        nxt += args.steps
    block_s = parallel.reduce_max_vector(block_s, rdev) if world > 1 else block_s        # a block takes as long as its slowest rank
    _, mean_pos = parallel.reduce_max_sum(0.0, float(np.mean(block_pos)), rdev) if world > 1 else (0.0, float(np.mean(block_pos)))
    order = np.argsort(block_s)
    med = int(order[len(order) // 2])             # the median block (for an even count the slower of the middle two)
    elapsed, tot_pos = block_s[med], (mean_pos if world > 1 else block_pos[med])
    fastest = float(np.min(block_s))
    kern_ms, kern_ms_min = float(np.median(kern)), float(np.min(kern))
    # the pyramid build alone, outside the timed steps (inside them it runs beside the walk)
    pev = [tuple(torch.cuda.Event(enable_timing=True) for _ in range(2)) for _ in range(30)]
    for k, (a, b) in enumerate(pev):
        a.record(stream_b)
        pyrs[0].rebuild_d(d_frames.data_ptr() + (k % nts) * fbytes, W, W * H)
        b.record(stream_b)
    torch.cuda.synchronize()
    issued["for"] = -1
    pyr_ms = float(np.median([a.elapsed_time(b) for a, b in pev]))
    pyr_bytes = nseq * (W * H + sum(pyr.level(i).w * pyr.level(i).h for i in range(1, len(SUBS))))

    # PCIe-inclusive figure (never `value`): the same step in its sequential form with the upload of the time-step's
    # frames from pinned host memory inside it
    with_h2d = None
    if rank == 0 and world == 1 and not args.no_side and not args.with_h2d:
        hp = torch.from_numpy(frames).pin_memory()
        p3 = api.ImagePyramid(ctx, None, SUBS, device_ptr=d_frames.data_ptr(), pitch=W, frame_stride=W * H, shape=(nseq, H, W))

        def h2d_step(k):
            t = k % nts
            d_frames[t].copy_(hp[t], non_blocking=True)
            p3.rebuild_d(d_frames.data_ptr() + t * fbytes, W, W * H)
            capi.check(ctx.handle, L.slam_hip_track_batch_d(ctx.handle, p3.handle, C.c_void_p(d_jobs[t].data_ptr()), njobs,
                                                            C.c_void_p(d_blob.data_ptr()), blob.size, C.c_void_p(d_out[t].data_ptr())))
        nh = max(4, min(args.steps, 20))
        ts_h = []
        for rep in range(4):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for k in range(nh):
                h2d_step(k)
            torch.cuda.synchronize()
            ts_h.append((time.perf_counter() - t0) / nh)
        ms_h = float(np.median(ts_h[1:])) * 1e3
        with_h2d = {"ms_per_step": ms_h, "value": float(np.mean(positions)) / (ms_h * 1e-3), "unit": "positions/s",
                    "h2d_bytes_per_step": int(fbytes), "h2d_gb_per_s_if_upload_alone": fbytes / (ms_h * 1e-3) / 1e9,
                    "note": "sequential step with the upload of its %d frames from pinned host memory inside it; not `value`" % nseq}
        p3.close()
        del hp

    side = {}
    if rank == 0 and world == 1 and not args.no_side:
        traffic = load_traffic()
        legs = set(args.legs.split(","))
        if not args.no_fast and "fast" in legs:
            side["fast"] = fast_leg(api, capi, ctx, stream, dev, args, traffic)
        if "c1" in legs:
            side["c1"] = c1_leg(api, capi, ctx, stream, dev, args, traffic)
        if "c3" in legs:
            side["c3"] = c3_leg(api, capi, ctx, stream, dev, args, traffic)
        if "prod" in legs:
            side["prod"] = prod_leg(api, capi, ctx, stream, dev, args, traffic)
        if "line" in legs:
            side["line_walk"] = line_leg(api, capi, ctx, stream, dev, args, traffic)

    if rank == 0:
        traffic = load_traffic()
        tr = traffic_of(traffic, "track_lean_kernel<false>@%d" % (8 * ((njobs + 7) // 8)), "track_kernel@%d" % min(njobs, 4096))
        ach = alg_bytes / (kern_ms * 1e-3) / 1e9
        line = {
            "metric": "NCC patch-match positions/sec (pyramid build + coarse-to-fine template search)",
            "value": tot_pos / elapsed,
            "unit": "positions/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "repeats": repeats, "block": "median of %d blocks of %d steps, each between barrier + synchronize" % (repeats, args.steps),
            "ms_per_step_min": fastest / args.steps * 1e3, "ms_per_step_max": float(np.max(block_s)) / args.steps * 1e3,
            "value_of_fastest_block": tot_pos / fastest,
            "calibration_block_ms_per_step": cal / args.steps * 1e3,
            "block_ms_per_step": [round(b / args.steps * 1e3, 4) for b in block_s[:64]],
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8", "data": "synthetic",
            "config": {"workload": "C2 shape (640x480, %d features/frame, %dx%d masked templates, pyramid {1,2,4}, stride 3, "
                                   "threshold 0.98), %d independent sequences per GPU x %d GPU(s) = %d sequences%s; a step = one "
                                   "time-step: one frame of every sequence; %d time-steps resident, cycled"
                                   % (NFEAT, TSIZE, TSIZE, nseq, world, nseq * world, " (C5)" if nseq * world == 4096 else "", nts),
                       "sequences_per_gpu": nseq, "sequences": nseq * world, "features_per_frame": NFEAT,
                       "positions_per_step_per_gpu": float(np.mean(positions)),
                       "frames_per_s": nseq * world * args.steps / elapsed, "tracked_fraction": tracked, "h2d_in_step": bool(args.with_h2d),
                       "pyramid_overlapped": bool(overlap),
                       "parity": None if args.no_cpu else "every rank compared a sample of each time-step's results with the oracle bit for bit",
                       "parallelism": "sequences sharded over %d GPU(s), no collective on the data path" % world},
            "roofline": {"bound": "hbm", "kernel": "track_lean_kernel+track_kernel", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": ach / HBM_PEAK_GBS, "frac_of_measured_copy_rate": ach / HBM_COPY_GBS, "traffic": tr,
                         "algorithmic_bytes_per_launch": alg_bytes, "kernel_ms": kern_ms, "kernel_ms_min": kern_ms_min, "jobs_per_launch": njobs,
                         "jobs_left_to_general_kernel": deferred.value,
                         "u8_mac_per_s": macs / (kern_ms * 1e-3),
                         "frac_of_packed_dot_peak": macs / (kern_ms * 1e-3) / DOT4_PEAK_MEASURED,
                         "frac_of_nominal_dot_peak": macs / (kern_ms * 1e-3) / DOT4_PEAK_NOMINAL,
                         "packed_dot_peak_measured": DOT4_PEAK_MEASURED, "packed_dot_peak_nominal": DOT4_PEAK_NOMINAL,
                         # the HBM fraction this workload would reach if nothing but its multiply-accumulates were issued
                         "hbm_frac_at_measured_dot_peak": alg_bytes / (macs / DOT4_PEAK_MEASURED) / 1e9 / HBM_PEAK_GBS,
                         "hbm_frac_at_nominal_dot_peak": alg_bytes / (macs / DOT4_PEAK_NOMINAL) / 1e9 / HBM_PEAK_GBS},
            "lds": lds_report(kern_ms, "track_lean_kernel<false>@%d" % (8 * ((njobs + 7) // 8)), "track_kernel@%d" % min(njobs, 4096)),
            "cpu_baseline": cpu,
            "with_h2d": with_h2d,
            # the other kernels and configurations BASELINE.json names, measured in the same run (outside the timed steps)
            "pyramid": {"kernel": "pyr_build_kernel", "ms": pyr_ms, "achieved": pyr_bytes / (pyr_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS,
                        "unit": "GB/s", "frac": pyr_bytes / (pyr_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                        "frac_of_measured_copy_rate": pyr_bytes / (pyr_ms * 1e-3) / 1e9 / HBM_COPY_GBS,
                        "algorithmic_bytes_per_launch": pyr_bytes},
        }
        line.update(side)
        print(json.dumps(line))
    for p in pyrs:
        p.close()
    if overlap:
        ctx_b.close()
    ctx.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
