#!/usr/bin/env python
"""Who runs beside the slow launches?  Reads a `rocprofv3 --kernel-trace` rocpd database (start / end per dispatch), takes the
launches of one kernel (default: the down-converter), and for its fastest and slowest tenth reports which other kernels
overlapped them and for what share of their duration; then the same for the slowest launches of every other kernel (the
message split, the block repair), with the dispatch-to-start ordering on their queue.

  python profiles/probe/tails.py <results.db> [kernel-substring]   
"""
import sqlite3
import sys


def short(n):
    n = n.split("(")[0]
    for k in ("fir_u8_direct", "fir_fmt_direct", "msk_demod", "blk_repair", "msg_split", "fill_random", "fir_u8_persist", "fir_u8_shared",
              "fir_u8_generic", "synth_iq", "read_probe", "copyBuffer", "fillBuffer"):
        if k in n:
            return k
    return n[:32]


def main(path, key="fir_u8_direct"):
    con = sqlite3.connect(path)
    rows = con.execute("select name, start, end, queue_id, stream_id, grid_x, workgroup_x from kernels order by start").fetchall()
    ks = [(short(n), s, e, q, st, g // max(1, w)) for n, s, e, q, st, g, w in rows]
    mine = [k for k in ks if key in k[0]]
    if not mine:
        mine = [k for k in ks if k[0].startswith("fir_")]
    if not mine:
        print("no launches of", key)
        return
    # steady state only: drop the first fifth (gate passes, warm-up)
    mine = mine[len(mine) // 5:]
    t_lo = mine[0][1]
    durs = sorted(e - s for _, s, e, *_ in mine)
    n = len(mine)
    print("# %s" % path)
    print("%s: %d launches (steady part)  min %.3f  p10 %.3f  median %.3f  p90 %.3f  max %.3f ms   max/min %.3f" % (
        mine[0][0], n, durs[0] / 1e6, durs[n // 10] / 1e6, durs[n // 2] / 1e6, durs[(9 * n) // 10] / 1e6, durs[-1] / 1e6, durs[-1] / durs[0]))

    def beside(s, e):
        """share of [s, e) during which each other kernel was running (sum over its overlapping launches, capped at 1)"""
        acc = {}
        for k, s2, e2, *_ in ks:
            if e2 <= s or s2 >= e or (s2 == s and e2 == e):
                continue
            acc[k] = acc.get(k, 0) + (min(e, e2) - max(s, s2))
        return {k: min(1.0, v / float(e - s)) for k, v in acc.items()}

    by_d = sorted(mine, key=lambda k: k[2] - k[1])
    for label, part in (("fastest tenth", by_d[: max(1, n // 10)]), ("middle tenth", by_d[n // 2 - max(1, n // 20): n // 2 + max(1, n // 20)]),
                        ("slowest tenth", by_d[-max(1, n // 10):])):
        tot = {}
        for _, s, e, *_ in part:
            for k, v in beside(s, e).items():
                tot[k] = tot.get(k, 0) + v
        mean_d = sum(e - s for _, s, e, *_ in part) / len(part)
        print("  %-14s mean %.3f ms; beside it (mean share of the launch's duration): %s" % (
            label, mean_d / 1e6, "  ".join("%s %.2f" % (k, v / len(part)) for k, v in sorted(tot.items(), key=lambda x: -x[1]) if k != mine[0][0])))
    # gap before each launch on its own queue (dispatch stalls show up as start - previous end)
    gaps = sorted(b[1] - a[2] for a, b in zip(mine[:-1], mine[1:]))
    if gaps:
        print("  gap between consecutive launches: median %.1f us  p90 %.1f us  max %.1f us" % (
            gaps[len(gaps) // 2] / 1e3, gaps[(9 * len(gaps)) // 10] / 1e3, gaps[-1] / 1e3))
    # the other kernels, steady part
    print("other kernels in the steady part:")
    others = {}
    for k in ks:
        if k[1] >= t_lo and k[0] != mine[0][0]:
            others.setdefault(k[0], []).append(k)
    for name, lst in sorted(others.items(), key=lambda x: -sum(k[2] - k[1] for k in x[1])):
        d = sorted(k[2] - k[1] for k in lst)
        m = len(d)
        line = "  %-16s %5d launches  min %8.1f  median %8.1f  p90 %8.1f  max %8.1f us  workgroups %s  queues %s" % (
            name, m, d[0] / 1e3, d[m // 2] / 1e3, d[(9 * m) // 10] / 1e3, d[-1] / 1e3,
            sorted(set(k[5] for k in lst))[:4], sorted(set(k[3] for k in lst)))
        print(line)
        if name in ("msg_split", "blk_repair"):
            worst = sorted(lst, key=lambda k: k[2] - k[1])[-max(1, m // 10):]
            tot = {}
            for _, s, e, *_ in worst:
                for k2, v in beside(s, e).items():
                    tot[k2] = tot.get(k2, 0) + v
            print("      slowest tenth (mean %.1f us) ran beside: %s" % (
                sum(k[2] - k[1] for k in worst) / len(worst) / 1e3,
                "  ".join("%s %.2f" % (k2, v / len(worst)) for k2, v in sorted(tot.items(), key=lambda x: -x[1]))))


if __name__ == "__main__":
    main(sys.argv[1], *(sys.argv[2:3]))
