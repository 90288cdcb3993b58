"""The lean walk kernel decides from sizes and LDS budget alone how it takes a level: whole, in blocks of cell rows and
chunks of template rows (its second build), or not at all.  The same arithmetic runs on the host (sizing of launches,
slam_hip_debug_walk_plan); its invariants need no GPU."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from slam_amd import capi  # noqa: E402


def _plan(tw, th, aw, ah, step, resized, budget, chunks):
    out = (C.c_int32 * 4)()
    assert capi.lib().slam_hip_debug_walk_plan(tw, th, aw, ah, step, int(resized), budget, int(chunks), out) == 0
    return tuple(out)


def test_known_shapes():
    # C2's full-resolution level (21x21 template, 17x17 window) fits the default 7680 bytes whole, in either build
    bh, R, nbytes, box = _plan(21, 21, 17, 17, 1, False, 7680, False)
    assert (bh, R, box) == (17, 21, 0) and 7000 < nbytes <= 7680
    assert _plan(21, 21, 17, 17, 1, False, 7680, True)[:2] == (17, 21)
    # C3's 31x31 templates need the 10240 the bench gives them
    assert _plan(31, 31, 17, 17, 1, False, 7680, False)[0] == 0
    assert _plan(31, 31, 17, 17, 1, False, 10240, False)[:2] == (17, 31)
    # the reference's live shape (81x81 descriptors, 24x24 window): never whole, the second build in blocks and chunks with box sums
    assert _plan(81, 81, 24, 24, 1, False, 24576, False)[0] == 0          # (whole it needs about 40 KB: 4 walks per CU)
    bh, R, nbytes, box = _plan(81, 81, 24, 24, 1, False, 12800, True)
    assert 0 < bh <= 24 and 4 <= R < 81 and nbytes <= 12800 and box == 2
    # its coarse level (27x27 resized, 14x14 window on the stride-3 lattice) is taken whole
    assert _plan(27, 27, 14, 14, 3, True, 12800, True)[:2] == (5, 27)
    # bad arguments
    out = (C.c_int32 * 4)()
    assert capi.lib().slam_hip_debug_walk_plan(0, 5, 5, 5, 1, 0, 8192, 1, out) == capi.EINVAL
    assert capi.lib().slam_hip_debug_walk_plan(5, 5, 5, 5, 1, 0, 8192, 1, None) == capi.EINVAL


def test_plans_fit_their_budget_and_grow_with_it():
    rng = np.random.default_rng(11)
    for _ in range(3000):
        tw, th = int(rng.integers(1, 130)), int(rng.integers(1, 130))
        aw, ah = int(rng.integers(1, 80)), int(rng.integers(1, 80))
        step = int(rng.choice([1, 1, 2, 3, 5]))
        resized = bool(rng.integers(0, 2))
        budget = int(rng.integers(64, 1024)) * 64
        rows_all = ah if step == 1 else (ah + step - 1) // step
        whole = _plan(tw, th, aw, ah, step, resized, budget, False)
        bh, R, nbytes, box = _plan(tw, th, aw, ah, step, resized, budget, True)
        if bh:
            assert nbytes <= budget
            assert min(3, rows_all) <= bh <= rows_all
            assert R == th or 2 <= R < th                  # (equal chunks: at least 4 rows before the rounding)
            assert box in (0, 2) and (box == 0 or (step == 1 and not resized and aw + tw - 1 <= 128))
            # more LDS never loses the level
            assert _plan(tw, th, aw, ah, step, resized, budget + 4096, True)[0] > 0
        if whole[0] and tw * th * ((aw + 3) // 4) < 2048 * 4:
            # what the plain build takes whole, the second build takes too (its tile is never larger)
            assert bh > 0
