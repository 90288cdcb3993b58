#!/usr/bin/env python3
"""Per dispatch of valu_rate2.bin: kernel, waves per SIMD (from the grid), duration, GRBM_GUI_ACTIVE cycles and the
clock they imply, VALU instructions and the cycles per wave instruction per SIMD at that clock.
usage: summarise_valu_pmc.py <rocprofv3 output dir>"""
import glob, os, sqlite3, sys
from collections import defaultdict


def main():
    src = sys.argv[1]
    want = ("k_add", "k_and", "k_max", "k_perm", "k_dot4", "k_lshlrev", "k_mad_u24")
    for db in sorted(glob.glob(os.path.join(src, "**", "*_results.db"), recursive=True)):
        con = sqlite3.connect(db)
        tabs = [r[0] for r in con.execute("select name from sqlite_master where type in ('table','view')")]
        cc = con.execute("select dispatch_id, kernel_name, grid_size, workgroup_size, counter_name, value from counters_collection").fetchall()
        by = defaultdict(dict)
        meta = {}
        for did, k, g, wg, c, v in cc:
            by[did][c] = by[did].get(c, 0.0) + v
            meta[did] = (k.split("(")[0], g // wg)
        dur = {}
        if "kernels" in tabs:
            for did, s, e in con.execute("select dispatch_id, start, end from kernels"):
                dur[did] = e - s
        print("%-12s %5s %9s %14s %9s %14s %10s %9s" % ("kernel", "w/SIMD", "dur_us", "GRBM_GUI_ACT", "clk_GHz", "SQ_INSTS_VALU", "cyc/inst", "VALUbusy"))
        seen = defaultdict(int)
        for did in sorted(by):
            k, blocks = meta[did]
            if k not in want:
                continue
            seen[(k, blocks)] += 1
            if seen[(k, blocks)] != 2:          # the second launch of each pair is the timed one
                continue
            w = blocks // 256
            c = by[did]
            d = dur.get(did)
            grbm = c.get("GRBM_GUI_ACTIVE", 0.0)
            insts = c.get("SQ_INSTS_VALU", 0.0)
            clk = grbm / d if d else float("nan")            # cycles per ns = GHz
            per = grbm / (insts / 1024.0) if insts else float("nan")      # graphics-clock cycles per wave instruction per SIMD
            busy = c.get("SQ_ACTIVE_INST_VALU", 0.0) / max(c.get("SQ_BUSY_CYCLES", 1.0), 1.0)
            print("%-12s %5d %9.1f %14.0f %9.3f %14.0f %10.2f %9.2f" % (k, w, (d or 0) / 1e3, grbm, clk, insts, per, busy))


if __name__ == "__main__":
    main()
