#!/usr/bin/env python3
"""Turn a rocprofv3 `--kernel-trace --stats` result (rocpd sqlite .db or *_kernel_stats.csv) into the
plain-text per-kernel summary committed under profiles/.   usage: summarize_rocprof.py <dir-or-file> [title]"""
import csv
import glob
import os
import sqlite3
import sys


def rows_from_db(path):
    c = sqlite3.connect(path)
    return [(r[0], r[1], r[2], r[3], r[4]) for r in   # rocpd top_kernels durations are in microseconds
            c.execute("select name,total_calls,total_duration,average,percentage from top_kernels")]


def rows_from_csv(path):
    out = []
    for r in csv.DictReader(open(path)):
        out.append((r["Name"], int(r["Calls"]), float(r["TotalDurationNs"]) / 1e3, float(r["AverageNs"]) / 1e3,
                    float(r["Percentage"])))
    return out


def main():
    target = sys.argv[1]
    title = sys.argv[2] if len(sys.argv) > 2 else target
    files = [target] if os.path.isfile(target) else \
        glob.glob(os.path.join(target, "**", "*_results.db"), recursive=True) + \
        glob.glob(os.path.join(target, "**", "*kernel_stats.csv"), recursive=True)
    if not files:
        raise SystemExit("no rocprofv3 result under %s" % target)
    f = files[0]
    rows = rows_from_db(f) if f.endswith(".db") else rows_from_csv(f)
    print("# %s" % title)
    print("# source: rocprofv3 --kernel-trace --stats (%s)" % os.path.basename(f))
    print("%-78s %7s %12s %12s %7s" % ("kernel", "calls", "total_us", "avg_us", "pct"))
    for name, calls, tot, avg, pct in rows:
        if pct < 0.05:
            continue
        short = name if len(name) <= 76 else name[:73] + "..."
        print("%-78s %7d %12.1f %12.2f %7.2f" % (short, calls, tot, avg, pct))


if __name__ == "__main__":
    main()
