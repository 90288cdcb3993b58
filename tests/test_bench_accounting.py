"""The arithmetic behind bench.py's units: positions and algorithmic bytes per feature of the C2 workload, and that
the counted positions are the cells the oracle's search visits."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def test_c2_positions_and_bytes_per_feature():
    lv = bench.level_cells_and_bytes()
    assert [l["level"] for l in lv] == [2, 1, 0]
    assert [l["tw"] for l in lv] == [5, 10, 21]                       # feature.cpp:130: Size(21/4, 21/4), (21/2, 21/2)
    assert [l["win"] for l in lv] == [11, 6, 17]                      # feature.cpp:134-146: 2*rad+1 rounded
    assert [l["cells"] for l in lv] == [16, 4, 289]                   # stride-3 lattices, stride 1 at full resolution
    assert sum(l["cells"] for l in lv) == 309
    # image tile under all candidate patches + template + mask + 20-byte result record, per job-level
    assert [l["bytes"] for l in lv] == [15 * 15 + 2 * 25 + 20, 15 * 15 + 2 * 100 + 20, 37 * 37 + 2 * 441 + 20]
    assert sum(l["bytes"] for l in lv) == 3011
    assert sum(l["macs"] for l in lv) == 2 * (16 * 25 + 4 * 100 + 289 * 441)


def test_counted_cells_are_the_cells_the_search_visits(oracle):
    """One interior feature of the bench workload: the oracle's score maps (cells other than -1 are visited) hold
    exactly the counted number of cells at every level."""
    rng = np.random.default_rng(1)
    frames, jobs, blob, truth = bench.build_time_steps(1, 1, 123, rng)
    frames, jobs = frames[0], jobs[0]
    levels = oracle.pyramid(frames[0], bench.SUBS)
    lv = {l["level"]: l for l in bench.level_cells_and_bytes()}
    j = jobs[NFEAT_MID]
    ts = bench.TSIZE
    proj = blob[j["proj_off"]:j["proj_off"] + ts * ts].reshape(ts, ts)
    mask = blob[j["mask_off"]:j["mask_off"] + ts * ts].reshape(ts, ts)
    tx, ty = float(j["px"]), float(j["py"])
    for i in (2, 1, 0):
        sub = float(bench.SUBS[i])
        tw = lv[i]["tw"]
        t = oracle.resize(proj, tw, tw) if tw != ts else proj
        m = (oracle.resize(mask, tw, tw) != 0).astype(np.uint8) if tw != ts else mask
        win = lv[i]["win"]
        rad = (win - 1) / 2.0
        area = (int(round(tx / sub - rad)), int(round(ty / sub - rad)), win, win)
        r = oracle.match(levels[i], t, m, area, None, 1 if i == 0 else bench.STRIDE, False, want_score=True)
        assert int((r["score"] != -1).sum()) == lv[i]["cells"], i
        tx, ty = r["x"] * sub, r["y"] * sub


NFEAT_MID = 27      # a feature away from the image border


def test_time_steps_share_templates_and_know_their_truth():
    """Two sequences, three time-steps: every time-step reuses the sequence's templates, frames differ by the
    integer shift the truth records, and the job's prediction lies within 4 pixels of it."""
    rng = np.random.default_rng(5)
    frames, jobs, blob, truth = bench.build_time_steps(2, 3, 77, rng)
    assert frames.shape == (3, 2, bench.H, bench.W) and len(jobs) == 3
    for t in range(3):
        j = jobs[t]
        assert len(j) == 2 * bench.NFEAT and set(j["frame"]) == {0, 1}
        assert np.array_equal(j["proj_off"], jobs[0]["proj_off"]) and np.array_equal(j["mask_off"], jobs[0]["mask_off"])
        assert (np.abs(j["px"] - truth[t][:, 0]) < 4).all() and (np.abs(j["py"] - truth[t][:, 1]) < 4).all()
        # the template of a feature is the frame content around its true position
        k = bench.NFEAT + NFEAT_MID
        ts = bench.TSIZE
        tx, ty = truth[t][k]
        cut = frames[t, 1][ty - ts // 2:ty + ts // 2 + 1, tx - ts // 2:tx + ts // 2 + 1]
        assert np.array_equal(cut.reshape(-1), blob[j["proj_off"][k]:j["proj_off"][k] + ts * ts])
