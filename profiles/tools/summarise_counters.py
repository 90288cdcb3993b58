#!/usr/bin/env python3
"""Per-kernel means of every counter in the rocprofv3 result databases under a directory (one sub-directory per
--pmc pass), per launch and per workgroup.  usage: summarise_counters.py <dir> [kernel-substring]"""
import glob
import os
import sqlite3
import sys


def main():
    src = sys.argv[1]
    want = sys.argv[2] if len(sys.argv) > 2 else ""
    print("%-34s %-24s %8s %16s %12s" % ("kernel", "counter", "launches", "mean/launch", "per wg"))
    for db in sorted(glob.glob(os.path.join(src, "**", "*_results.db"), recursive=True)):
        con = sqlite3.connect(db)
        try:
            rows = con.execute("select kernel_name, counter_name, count(*), avg(value), avg(grid_size / workgroup_size) "
                               "from counters_collection group by kernel_name, counter_name order by kernel_name, counter_name").fetchall()
        except sqlite3.Error as e:
            print("# %s: %s" % (db, e))
            continue
        for k, c, n, v, wg in rows:
            k = k.replace("(anonymous namespace)::", "").split("(")[0]
            if want not in k or k.startswith("void at::") or "rocclr" in k:
                continue
            print("%-34s %-24s %8d %16.0f %12.1f" % (k[:34], c, n, v, v / max(wg, 1)))


if __name__ == "__main__":
    main()
