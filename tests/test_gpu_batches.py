"""Batches over several frames: every job kind names its frame, and jobs of different frames share one launch.  Each
frame of the batch is shifted differently, so a job that read the wrong frame would find its feature elsewhere."""
import numpy as np
import pytest

from slam_amd import api, synth
from test_gpu_lines import _line_jobs

pytestmark = pytest.mark.gpu


def _bits(v):
    return int(np.float32(v).view(np.uint32))


def test_line_window_and_projection_jobs_across_frames(ctx, oracle):
    w, h, subs = 320, 240, [1.0, 2.0]
    shifts = [(4, -3), (-5, 2), (1, 6)]
    f0 = synth.frame(w, h, 51)
    frames = np.stack([synth.frame(w, h, 51, s) for s in shifts])
    rng = np.random.default_rng(52)
    line_jobs, win_jobs = [], []
    for f, s in enumerate(shifts):
        for j in _line_jobs(rng, w, h, subs, 4, 12, f0, s):
            line_jobs.append(dict(j, frame=f))
        for (cx, cy) in synth.feature_grid(w, h, 6, 50, rng):
            proj = synth.cut_template(f0, cx, cy, 15, 15)
            px, py = cx - s[0] + 2.5, cy - s[1] - 1.25
            cov = np.array([40.0 + 5 * f, 6.0, 30.0], np.float32)
            win_jobs.append(dict(frame=f, proj=proj, proj_mask=np.ones_like(proj), px=px, py=py, threshold=0.9, stride=2,
                                 window=api.ellipse_setup(px, py, cov, 9.0, 1.0 / subs[-1]), cov=cov, truth=(cx - s[0], cy - s[1])))
    order = rng.permutation(len(line_jobs))                      # frames interleaved within the launch
    line_jobs = [line_jobs[i] for i in order]
    order = rng.permutation(len(win_jobs))
    win_jobs = [win_jobs[i] for i in order]
    pyr = api.ImagePyramid(ctx, frames, subs)
    try:
        got_l = api.match_lines_batch(ctx, pyr, line_jobs)
        got_w = api.track_batch(ctx, pyr, win_jobs)
    finally:
        pyr.close()
    levels = [oracle.pyramid(frames[f], subs) for f in range(len(shifts))]
    refined = 0
    for j, g in zip(line_jobs, got_l):
        am = oracle.disc_mask(j["area"][2], j["area"][3], j["circles"])
        r = oracle.line_track(levels[j["frame"]], subs, j["descriptor"], j["area"], am, j["threshold"], j["stride"])
        assert (g["status"], g["x"], g["y"], g["levels_run"], _bits(g["response"])) == (0, r["x"], r["y"], r["levels_run"], _bits(r["response"]))
        refined += r["levels_run"] == 2
    assert refined >= len(line_jobs) // 2
    found = 0
    for j, g in zip(win_jobs, got_w):
        o = oracle.ellipse_window(j["px"], j["py"], j["cov"], 9.0, 1.0 / subs[-1])
        r = oracle.track(levels[j["frame"]], subs, j["proj"], j["proj_mask"], j["px"], j["py"], j["threshold"], j["stride"],
                         coarse_area=o["area"], coarse_mask=o["mask"])
        assert (g["status"], g["x"], g["y"], g["levels_run"], _bits(g["response"])) == (0, r["x"], r["y"], r["levels_run"], _bits(r["response"]))
        found += (g["x"], g["y"]) == j["truth"] and g["response"] > 0.9
    assert found >= len(win_jobs) // 2
