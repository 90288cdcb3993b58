#!/usr/bin/env python3
"""Per-launch means of the SQ counters collected by profiles/tools/sq_counters.sh, per kernel and per job
(65 536 jobs per launch).  usage: summarise_sq.py gpurun_out/prof_sq <tag>"""
import glob
import os
import sqlite3
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    src, tag = sys.argv[1], sys.argv[2]
    lines = ["rocprofv3 --pmc <SQ counters> -- python3 profiles/tools/track_probe.py one   (1024 frames x 64 features, threshold -10: all three levels run)",
             "%-28s %-24s %10s %16s %14s" % ("kernel", "counter", "launches", "mean/launch", "per job")]
    for db in sorted(glob.glob(os.path.join(src, "*", "*_results.db"))):
        con = sqlite3.connect(db)
        rows = con.execute("select kernel_name, counter_name, count(*), avg(value) from counters_collection "
                           "group by kernel_name, counter_name order by kernel_name, counter_name").fetchall()
        for k, c, n, v in rows:
            k = k.replace("(anonymous namespace)::", "").split("(")[0]
            if "track_kernel" not in k:
                continue
            lines.append("%-28s %-24s %10d %16.0f %14.1f" % (k[:28], c, n, v, v / 65536.0))
    text = "\n".join(lines)
    with open(os.path.join(ROOT, "profiles", tag + "_track_sq_counters.txt"), "w") as fh:
        fh.write(text + "\n")
    print(text)


if __name__ == "__main__":
    main()
