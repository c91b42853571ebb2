#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd database (bench_results.db) as text.

usage: rocpd_summary.py <kernel-trace db> [--pmc <db> ...]
 * kernel-trace db: per (kernel, grid) launch count / total / average / min / max duration (us)
 * --pmc db: per (kernel, grid) average of every collected counter per launch
"""
import sqlite3
import sys


def short(name, n=110):
    name = name.replace("unsigned long long", "u64").replace("unsigned int", "u32")
    return name if len(name) <= n else name[:n - 3] + "..."


def kernel_table(path):
    db = sqlite3.connect(path)
    rows = db.execute(
        "select name, grid_x, workgroup_x, count(*), sum(duration), avg(duration), min(duration), max(duration), "
        "max(vgpr_count), max(lds_size) from kernels group by name, grid_x, workgroup_x order by sum(duration) desc").fetchall()
    total = sum(r[4] for r in rows) or 1
    out = ["%-112s %10s %6s %6s %12s %10s %10s %10s %6s %5s %8s" % ("kernel", "grid", "wg", "calls", "total_us", "avg_us", "min_us", "max_us", "pct", "vgpr", "lds")]
    for name, gx, wx, cnt, tot, avg, mn, mx, vg, lds in rows:
        out.append("%-112s %10d %6d %6d %12.1f %10.2f %10.2f %10.2f %6.2f %5d %8d" % (
            short(name), gx, wx, cnt, tot / 1e3, avg / 1e3, mn / 1e3, mx / 1e3, 100.0 * tot / total, vg or 0, lds or 0))
    return "\n".join(out)


def pmc_table(path):
    db = sqlite3.connect(path)
    rows = db.execute(
        "select kernel_name, grid_size_x, counter_name, count(*), avg(value), avg(duration) from counters_collection "
        "group by kernel_name, grid_size_x, counter_name order by sum(duration) desc").fetchall()
    out = ["%-112s %10s %-14s %6s %16s %10s" % ("kernel", "grid", "counter", "calls", "avg_value/launch", "avg_us")]
    for name, gx, cname, cnt, val, dur in rows:
        out.append("%-112s %10d %-14s %6d %16.1f %10.2f" % (short(name), gx, cname, cnt, val, dur / 1e3))
    return "\n".join(out)


def main():
    args = sys.argv[1:]
    if not args:
        print(__doc__)
        return 1
    i = 0
    while i < len(args):
        if args[i] == "--pmc":
            print("\n== PMC: %s ==" % args[i + 1])
            print(pmc_table(args[i + 1]))
            i += 2
        else:
            print("== kernel trace: %s ==" % args[i])
            print(kernel_table(args[i]))
            i += 1
    return 0


if __name__ == "__main__":
    sys.exit(main())
