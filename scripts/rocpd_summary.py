#!/usr/bin/env python
"""Summarise a rocprofv3 rocpd sqlite database (bench_results.db) as text:
per-kernel launch count / total / average duration (the `--stats` view) and, if the run
collected PMC counters, per-kernel average counter values."""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(.*", "", name)
    return name.split("::")[-1][:60]


def main(path):
    db = sqlite3.connect(path)
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    print("# %s" % path)
    rows = cur.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                       "from kernels group by name order by sum(end-start) desc").fetchall()
    tot = sum(r[2] for r in rows) or 1
    print("%-62s %8s %12s %12s %12s %12s %7s" % ("kernel", "calls", "total_ms", "avg_ms", "min_ms", "max_ms", "pct"))
    for n, c, s, a, mn, mx in rows:
        print("%-62s %8d %12.3f %12.3f %12.3f %12.3f %6.1f%%" % (short(n), c, s / 1e6, a / 1e6, mn / 1e6, mx / 1e6, 100.0 * s / tot))
    try:
        pm = cur.execute("select kernel_name, counter_name, count(*), avg(value), sum(value) from counters_collection "
                         "group by kernel_name, counter_name order by kernel_name").fetchall()
    except Exception as e:
        pm = []
    if pm:
        print("\n%-62s %-14s %8s %16s" % ("kernel", "counter", "calls", "avg_value"))
        for kn, cn, c, a, s in pm:
            print("%-62s %-14s %8d %16.1f" % (short(kn), cn, c, a))


if __name__ == "__main__":
    for p in sys.argv[1:]:
        main(p)
        print()
