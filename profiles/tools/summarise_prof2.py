#!/usr/bin/env python3
"""Turns the rocprofv3 result databases written by profiles/tools/profile_bench.sh into the text summaries kept under
profiles/ and the HBM-traffic file bench.py reads.  Kernels are told apart by name AND workgroup count, because the
same kernel serves several legs of the benchmark (C2 and C3 walks, ...).

    python profiles/tools/summarise_prof2.py gpurun_out/prof_r02 r02 [sequences_per_gpu]

Outputs: profiles/<tag>_kernel_stats.txt, profiles/<tag>_pmc_traffic.txt, profiles/<tag>_traffic.json
"""
import json
import os
import sqlite3
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CALIB_BYTES = 512 << 20


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    return name.split("(")[0][:48]


def kernel_stats(db):
    rows = db.execute("select name, grid_x * grid_y * grid_z / (workgroup_x * workgroup_y * workgroup_z), count(*), sum(duration), avg(duration), "
                      "min(duration), max(duration) from kernels group by name, 2 order by sum(duration) desc").fetchall()
    tot = sum(r[3] for r in rows) or 1
    out = ["%-48s %9s %6s %12s %12s %12s %12s %7s" % ("kernel", "workgroups", "calls", "total_us", "avg_us", "min_us", "max_us", "pct")]
    avg = {}
    for n, wg, c, s, a, lo, hi in rows:
        if short(n).startswith("at::") or "rocclr" in n:
            continue
        out.append("%-48s %9d %6d %12.1f %12.2f %12.2f %12.2f %6.2f%%" % (short(n), wg, c, s / 1e3, a / 1e3, lo / 1e3, hi / 1e3, 100.0 * s / tot))
        avg["%s@%d" % (short(n), wg)] = a / 1e3
    return "\n".join(out), avg


def counter_by_kernel(path, counter):
    """(kernel@workgroups) -> (launches, mean KB per launch) for one counter."""
    db = sqlite3.connect(path)
    rows = db.execute("select kernel_name, grid_size / workgroup_size, count(*), avg(value) from counters_collection where counter_name=? "
                      "group by kernel_name, 2", (counter,)).fetchall()
    return {"%s@%d" % (short(n), wg): (c, v) for n, wg, c, v in rows if not short(n).startswith("at::") and "rocclr" not in n}


def main():
    src, tag = sys.argv[1], sys.argv[2]
    nseq = int(sys.argv[3]) if len(sys.argv) > 3 else 512
    prof = os.path.join(ROOT, "profiles")
    text, avg = kernel_stats(sqlite3.connect(os.path.join(src, "stats", "stats_results.db")))
    with open(os.path.join(prof, tag + "_kernel_stats.txt"), "w") as fh:
        fh.write("rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 20 --warmup 3 --repeats 5 --no-cpu --no-fast-sparse   (1 GPU, %d sequences per step)\n" % nseq)
        fh.write(text + "\n")
    print(text)
    fetch = counter_by_kernel(os.path.join(src, "fetch", "fetch_results.db"), "FETCH_SIZE")
    write = counter_by_kernel(os.path.join(src, "write", "write_results.db"), "WRITE_SIZE")
    cf = counter_by_kernel(os.path.join(src, "calib_fetch", "calib_results.db"), "FETCH_SIZE")
    cw = counter_by_kernel(os.path.join(src, "calib_write", "calib_results.db"), "WRITE_SIZE")
    lines = ["rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python3 bench.py --steps 3 --warmup 1 --no-cpu --no-fast-sparse",
             "counter values are KB per launch (mean over launches); read bytes = FETCH_SIZE x 1024 x correction", ""]
    lines.append("calibration (profiles/tools/pmc_calib.hip; each read kernel reads %d bytes once):" % CALIB_BYTES)
    factor = {}
    for k, (c, v) in sorted(cf.items()):
        known = CALIB_BYTES
        if k.startswith("calib_read_rows"):
            known = CALIB_BYTES // 640 * 128     # 40-byte row segments: one 128-byte line each
        if k.startswith("calib_read"):
            factor[k.split("@")[0]] = known / (v * 1024.0)
            lines.append("  %-32s FETCH_SIZE %12.1f KB  -> bytes/counted-byte = %.3f (known %d B)" % (k, v, known / (v * 1024.0), known))
    for k, (c, v) in sorted(cw.items()):
        if k.startswith("calib_write"):
            lines.append("  %-32s WRITE_SIZE %12.1f KB  -> bytes/counted-byte = %.3f" % (k, v, CALIB_BYTES / (v * 1024.0)))
    corr = factor.get("calib_read_b32", 2.0)
    lines += ["", "correction used for dword-per-lane reads: x%.3f" % corr, ""]
    lines.append("%-58s %8s %14s %14s %16s" % ("kernel@workgroups", "launches", "FETCH_SIZE_KB", "WRITE_SIZE_KB", "HBM_bytes/launch"))
    out = {"sequences_per_gpu": nseq, "avg_kernel_us": avg, "kernels": {}, "lds": {}}
    # LDS pass: LDS-array cycles (all CUs summed), the conflict cycles among them, and GRBM_GUI_ACTIVE (graphics-clock
    # cycles while the kernel ran, summed over the 8 XCDs) per launch.  busy = array cycles / (cycles per XCD x 256 CUs);
    # bytes <= conflict-free array cycles x 256 B (the array's width): an upper bound on what the kernel moved through it.
    lds_db = os.path.join(src, "lds", "lds_results.db")
    if os.path.exists(lds_db):
        idx = counter_by_kernel(lds_db, "SQ_LDS_IDX_ACTIVE")
        bc = counter_by_kernel(lds_db, "SQ_LDS_BANK_CONFLICT")
        ins = counter_by_kernel(lds_db, "SQ_INSTS_LDS")
        grbm = counter_by_kernel(lds_db, "GRBM_GUI_ACTIVE")
        ll = ["rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS GRBM_GUI_ACTIVE -- python3 bench.py --steps 3 --warmup 1 --repeats 2 --no-cpu --no-fast-sparse",
              "per launch, means over launches; array cycles are summed over the 256 CUs, GRBM cycles over the 8 XCDs", "",
              "%-44s %8s %14s %14s %12s %12s %8s %10s" % ("kernel@workgroups", "launches", "array_cycles", "conflict_cyc", "lds_insts", "cyc_per_xcd", "busy", "conflicts")]
        for k in sorted(idx):
            a, c, n, g = idx[k][1], bc.get(k, (0, 0.0))[1], ins.get(k, (0, 0.0))[1], grbm.get(k, (0, 0.0))[1] / 8.0
            busy = a / (g * 256.0) if g else 0.0
            out["lds"][k] = {"array_cycles": a, "conflict_cycles": c, "lds_insts": n, "cycles_per_xcd": g, "array_busy": busy,
                             "bytes_upper_bound": (a - c) * 256.0}
            ll.append("%-44s %8d %14.0f %14.0f %12.0f %12.0f %7.1f%% %9.1f%%" % (k[:44], idx[k][0], a, c, n, g, 100 * busy, 100 * c / a if a else 0))
        with open(os.path.join(prof, tag + "_lds.txt"), "w") as fh:
            fh.write("\n".join(ll) + "\n")
        print("\n".join(ll))
    for k in sorted(set(fetch) | set(write)):
        f = fetch.get(k, (0, 0.0))
        w = write.get(k, (0, 0.0))
        hbm = f[1] * 1024.0 * corr + w[1] * 1024.0
        out["kernels"][k] = {"fetch_size_kb": f[1], "write_size_kb": w[1], "read_correction": corr, "hbm_bytes_per_launch": hbm}
        lines.append("%-58s %8d %14.1f %14.1f %16.0f" % (k[:58], f[0], f[1], w[1], hbm))
    with open(os.path.join(prof, tag + "_pmc_traffic.txt"), "w") as fh:
        fh.write("\n".join(lines) + "\n")
    print("\n".join(lines))
    with open(os.path.join(prof, tag + "_traffic.json"), "w") as fh:
        json.dump(out, fh, indent=1)


if __name__ == "__main__":
    main()
