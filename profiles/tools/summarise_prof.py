#!/usr/bin/env python3
"""Turns the rocprofv3 result databases written by profile_bench.sh into the text summaries kept
under profiles/ and the HBM-traffic file bench.py reads.

    python profiles/tools/summarise_prof.py gpurun_out/prof_r01 r01 [frames_per_gpu]

Outputs: profiles/<tag>_kernel_stats.txt, profiles/<tag>_pmc_traffic.txt, profiles/<tag>_track_traffic.json
"""
import json
import os
import sqlite3
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CALIB_BYTES = 512 << 20


def short(name):
    name = name.replace("(anonymous namespace)::", "")
    return name.split("(")[0][:60]


def kernel_stats(db):
    rows = db.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration) "
                      "from kernels group by name order by sum(duration) desc").fetchall()
    tot = sum(r[2] for r in rows) or 1
    out = ["%-60s %6s %12s %12s %12s %12s %7s" % ("kernel", "calls", "total_us", "avg_us", "min_us", "max_us", "pct")]
    for n, c, s, a, lo, hi in rows:
        out.append("%-60s %6d %12.1f %12.2f %12.2f %12.2f %6.2f%%" % (short(n), c, s / 1e3, a / 1e3, lo / 1e3, hi / 1e3, 100.0 * s / tot))
    return "\n".join(out), {short(n): a / 1e3 for n, c, s, a, lo, hi in rows}


def counter_by_kernel(path, counter):
    """kernel -> (launches, mean KB per launch) for one counter."""
    db = sqlite3.connect(path)
    rows = db.execute("select kernel_name, count(*), avg(value) from counters_collection where counter_name=? "
                      "group by kernel_name", (counter,)).fetchall()
    return {short(n): (c, v) for n, c, v in rows}


def main():
    if sys.argv[1] == "--side":        # kernel table of a `rocprofv3 --kernel-trace --stats -- python3 profiles/tools/side_probes.py` run
        text, _ = kernel_stats(sqlite3.connect(sys.argv[2]))
        with open(os.path.join(ROOT, "profiles", sys.argv[3] + "_side_kernel_stats.txt"), "w") as fh:
            fh.write("rocprofv3 --kernel-trace --stats -- python3 profiles/tools/side_probes.py   (1 GPU)\n" + text + "\n")
        print(text)
        return
    src, tag = sys.argv[1], sys.argv[2]
    frames = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
    prof = os.path.join(ROOT, "profiles")
    text, avg = kernel_stats(sqlite3.connect(os.path.join(src, "stats", "stats_results.db")))
    with open(os.path.join(prof, tag + "_kernel_stats.txt"), "w") as fh:
        fh.write("rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 20 --warmup 3 --no-cpu   (1 GPU, %d frames/step)\n" % frames)
        fh.write(text + "\n")
    print(text)

    fetch = counter_by_kernel(os.path.join(src, "fetch", "fetch_results.db"), "FETCH_SIZE")
    write = counter_by_kernel(os.path.join(src, "write", "write_results.db"), "WRITE_SIZE")
    cf = counter_by_kernel(os.path.join(src, "calib_fetch", "calib_results.db"), "FETCH_SIZE")
    cw = counter_by_kernel(os.path.join(src, "calib_write", "calib_results.db"), "WRITE_SIZE")
    lines = ["rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python3 bench.py --steps 3 --warmup 1 --no-cpu",
             "counter values are KB per launch (mean over launches); read bytes = FETCH_SIZE x 1024 x correction", ""]
    lines.append("calibration (profiles/tools/pmc_calib.hip; each read kernel reads %d bytes once):" % CALIB_BYTES)
    factor = {}
    for k, (c, v) in sorted(cf.items()):
        known = CALIB_BYTES
        if k.startswith("calib_read_rows"):
            known = CALIB_BYTES // 640 * 128     # 40-byte row segments: one 128-byte line each
        if k.startswith("calib_read"):
            factor[k] = known / (v * 1024.0)
            lines.append("  %-24s FETCH_SIZE %12.1f KB  -> bytes/counted-byte = %.3f (known %d B)" % (k, v, factor[k], known))
    for k, (c, v) in sorted(cw.items()):
        if k.startswith("calib_write"):
            lines.append("  %-24s WRITE_SIZE %12.1f KB  -> bytes/counted-byte = %.3f" % (k, v, CALIB_BYTES / (v * 1024.0)))
    corr = factor.get("calib_read_b32", 2.0)
    lines += ["", "correction used for dword-per-lane reads: x%.3f" % corr, ""]
    lines.append("%-40s %8s %14s %14s %16s" % ("kernel", "launches", "FETCH_SIZE_KB", "WRITE_SIZE_KB", "HBM_bytes/launch"))
    result = {}
    for k in sorted(set(fetch) | set(write)):
        f = fetch.get(k, (0, 0.0))
        w = write.get(k, (0, 0.0))
        hbm = f[1] * 1024.0 * corr + w[1] * 1024.0
        result[k] = {"fetch_size_kb": f[1], "write_size_kb": w[1], "read_correction": corr, "hbm_bytes_per_launch": hbm}
        lines.append("%-40s %8d %14.1f %14.1f %16.0f" % (k[:40], f[0], f[1], w[1], hbm))
    with open(os.path.join(prof, tag + "_pmc_traffic.txt"), "w") as fh:
        fh.write("\n".join(lines) + "\n")
    print("\n".join(lines))
    out = {"frames_per_gpu": frames, "avg_kernel_us": avg}
    for k, v in result.items():
        out[k] = v
    with open(os.path.join(prof, tag + "_track_traffic.json"), "w") as fh:
        json.dump(out, fh, indent=1)


if __name__ == "__main__":
    main()
