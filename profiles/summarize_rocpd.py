#!/usr/bin/env python
"""Turns rocprofv3's rocpd SQLite output (ROCm 7.2 default) into the plain-text summaries that are
committed under profiles/:  per-kernel call count / total / average / min / max duration, and the
per-kernel mean of every collected PMC counter.
usage: summarize_rocpd.py <results.db> [more.db ...]"""
import sqlite3
import sys


def summarize(path):
    con = sqlite3.connect(path)
    cur = con.cursor()
    out = ["# " + path]
    rows = cur.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration) "
                       "from kernels group by name order by sum(duration) desc").fetchall()
    tot = sum(r[2] for r in rows) or 1
    out.append("%-64s %8s %14s %12s %12s %12s %7s" % ("kernel", "calls", "total_ns", "avg_ns", "min_ns", "max_ns", "pct"))
    for n, c, s, a, mn, mx in rows:
        out.append("%-64s %8d %14d %12.0f %12d %12d %6.2f%%" % (n[:64], c, s, a, mn, mx, 100.0 * s / tot))
    try:
        pm = cur.execute("select k.name, p.counter_name, count(*), avg(p.counter_value), min(p.counter_value), max(p.counter_value) "
                         "from pmc_events p join kernels k on k.dispatch_id = p.dispatch_id "
                         "group by k.name, p.counter_name order by k.name").fetchall()
    except sqlite3.Error:
        pm = []
    if pm:
        out.append("")
        out.append("%-64s %-16s %8s %18s %18s %18s" % ("kernel", "counter", "n", "avg", "min", "max"))
        for n, cn, c, a, mn, mx in pm:
            out.append("%-64s %-16s %8d %18.1f %18.1f %18.1f" % (n[:64], cn, c, a, mn, mx))
    return "\n".join(out)


if __name__ == "__main__":
    for p in sys.argv[1:]:
        print(summarize(p))
        print()
