#!/usr/bin/env python3
"""Summarises a rocprofv3 rocpd SQLite result (kernel trace) into the per-kernel stats table that
`rocprofv3 --kernel-trace --stats` reports.  Usage: python tools/rocpd_summary.py results.db [> profiles/x.txt]"""
import sqlite3
import sys


def main(path):
    db = sqlite3.connect(path)
    rows = db.execute(
        "select name, count(*), sum(duration), avg(duration), min(duration), max(duration), max(vgpr_count), max(accum_vgpr_count), "
        "max(sgpr_count), max(lds_size), max(scratch_size), max(grid_x), max(workgroup_x) from kernels group by name order by sum(duration) desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    print("# source: %s (rocprofv3 --kernel-trace --stats, rocpd format); durations in ns" % path)
    print("%-62s %6s %14s %12s %12s %12s %6s %5s %5s %5s %7s %8s %10s %5s" % (
        "kernel", "calls", "total_ns", "avg_ns", "min_ns", "max_ns", "pct", "vgpr", "agpr", "sgpr", "lds_B", "scratch", "grid_x", "wg_x"))
    for r in rows:
        name = r[0] if len(r[0]) <= 60 else r[0][:57] + "..."
        print("%-62s %6d %14d %12.0f %12d %12d %6.2f %5d %5d %5d %7d %8d %10d %5d" % (
            name, r[1], r[2], r[3], r[4], r[5], 100.0 * r[2] / total, r[6] or 0, r[7] or 0, r[8] or 0, r[9] or 0, r[10] or 0, r[11] or 0, r[12] or 0))


if __name__ == "__main__":
    main(sys.argv[1])
