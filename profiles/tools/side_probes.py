#!/usr/bin/env python3
"""Times the two HBM-bound kernels of the path on their own, with inputs resident in HBM:
  - the pyramid build (C2: 1024 frames 640x480 {1,2,4}; C3: 256 frames 1280x720 {1,2,4,8})
  - FAST-9-16 + NMS (C4: 64 frames 1920x1080)
and prints the achieved fraction of the 8 TB/s HBM peak on the algorithmic bytes.
Diagnostic tool (the contract benchmark is bench.py)."""
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def timed(stream, fn, steps=20, warm=3):
    import torch
    for _ in range(warm):
        fn()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    for a, b in evs:
        a.record(stream); fn(); b.record(stream)
    torch.cuda.synchronize()
    return float(np.median([a.elapsed_time(b) for a, b in evs]))


def main():
    import torch
    from slam_amd import api, capi, synth
    dev = torch.device("cuda", 0)
    stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(stream)
    ctx = api.Context(0, stream=stream.cuda_stream)
    L = capi.lib()
    out = {}

    for name, (w, h, subs, nfr) in {"pyramid_c2": (640, 480, [1.0, 2.0, 4.0], 1024),
                                    "pyramid_c3": (1280, 720, [1.0, 2.0, 4.0, 8.0], 256),
                                    "pyramid_ref_1_3": (640, 480, [1.0, 3.0], 1024)}.items():
        base = synth.frame(w, h, 3)
        d = torch.from_numpy(base).to(dev).unsqueeze(0).repeat(nfr, 1, 1).contiguous()
        pyr = api.ImagePyramid(ctx, None, subs, device_ptr=d.data_ptr(), pitch=w, frame_stride=w * h, shape=(nfr, h, w))
        ms = timed(stream, lambda: pyr.rebuild_d(d.data_ptr(), w, w * h))
        alg = w * h + sum(pyr.level(i).w * pyr.level(i).h for i in range(1, len(subs)))
        gbs = alg * nfr / (ms * 1e-3) / 1e9
        out[name] = {"ms": ms, "frames": nfr, "alg_bytes_per_frame": alg, "GB/s": gbs, "frac_hbm": gbs / 8000.0,
                     "Mpx/s": w * h * nfr / (ms * 1e-3) / 1e6}
        pyr.close()
        del d

    w, h, nfr, cap = 1920, 1080, 64, 65536
    frames = np.stack([synth.frame(w, h, 100 + i) for i in range(4)])
    d = torch.from_numpy(frames).to(dev).repeat(nfr // 4, 1, 1).contiguous()
    sb = L.slam_hip_fast_scratch_bytes(w, h, nfr, cap)
    d_scr = torch.zeros(sb, dtype=torch.uint8, device=dev)
    d_out = torch.zeros(nfr * cap * 12, dtype=torch.uint8, device=dev)
    d_cnt = torch.zeros(nfr, dtype=torch.int32, device=dev)
    for thr in (10, 40):
        def go():
            capi.check(ctx.handle, L.slam_hip_fast_detect_d(ctx.handle, C.c_void_p(d.data_ptr()), w, h, w, w * h, nfr, thr, 1,
                                                            C.c_void_p(d_out.data_ptr()), cap, C.c_void_p(d_cnt.data_ptr()),
                                                            C.c_void_p(d_scr.data_ptr()), sb))
        ms = timed(stream, go)
        cnt = d_cnt.cpu().numpy()
        alg = w * h * nfr + 12 * int(np.minimum(cnt, cap).sum())
        gbs = alg / (ms * 1e-3) / 1e9
        out["fast_c4_thr%d" % thr] = {"ms": ms, "frames": nfr, "mean_corners_per_frame": float(cnt.mean()), "GB/s": gbs,
                                     "frac_hbm": gbs / 8000.0, "Gpx/s": w * h * nfr / (ms * 1e-3) / 1e9}
    print(json.dumps(out, indent=1))
    ctx.close()


if __name__ == "__main__":
    main()
