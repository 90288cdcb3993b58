#!/usr/bin/env python3
"""Times slam_hip_fast_detect_d on the C4 shape (64 frames 1920x1080) for one library build
(SLAM_HIP_LIB selects it).  Diagnostic tool."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    import torch
    from slam_amd import api, capi, synth
    dev = torch.device("cuda", 0)
    stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(stream)
    ctx = api.Context(0, stream=stream.cuda_stream)
    L = capi.lib()
    w, h, nfr, cap = 1920, 1080, 64, 65536
    frames = np.stack([synth.frame(w, h, 100 + i) for i in range(4)])
    d = torch.from_numpy(frames).to(dev).repeat(nfr // 4, 1, 1).contiguous()
    sb = L.slam_hip_fast_scratch_bytes(w, h, nfr, cap)
    d_scr = torch.zeros(sb, dtype=torch.uint8, device=dev)
    d_out = torch.zeros(nfr * cap * 12, dtype=torch.uint8, device=dev)
    d_cnt = torch.zeros(nfr, dtype=torch.int32, device=dev)
    for thr in (10, 40, 80):
        def go():
            capi.check(ctx.handle, L.slam_hip_fast_detect_d(ctx.handle, C.c_void_p(d.data_ptr()), w, h, w, w * h, nfr, thr, 1,
                                                            C.c_void_p(d_out.data_ptr()), cap, C.c_void_p(d_cnt.data_ptr()),
                                                            C.c_void_p(d_scr.data_ptr()), sb))
        for _ in range(40):
            go()
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(40)]
        for a, b in evs:
            a.record(stream); go(); b.record(stream)
        torch.cuda.synchronize()
        ms = float(np.median([a.elapsed_time(b) for a, b in evs]))
        print("%s thr=%d: %.3f ms, %.1f Gpx/s, corners/frame %.0f" % (os.environ.get("SLAM_HIP_LIB", "default"), thr, ms,
              w * h * nfr / ms / 1e6, d_cnt.cpu().numpy().mean()), flush=True)
    ctx.close()


if __name__ == "__main__":
    main()
