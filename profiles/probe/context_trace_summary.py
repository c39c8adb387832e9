"""Summarises a `rocprofv3 --kernel-trace` CSV of profiles/probe/context_probe.py: the dispatch sequence is cut at the
marker launches (fill_random), and per turn (= one context timed for a while) it prints, per kernel, the HSA queue(s) it
was dispatched on, the launch count, mean duration -- and the overlap of the two stages.

  python profiles/probe/context_trace_summary.py <dir with *_kernel_trace.csv> [contexts]
"""
import csv
import glob
import os
import sys

d = sys.argv[1]
nctx = int(sys.argv[2]) if len(sys.argv) > 2 else 4
files = sorted(glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True))
if not files:
    print("no kernel trace under", d)
    sys.exit(0)
rows = []
for f in files:
    with open(f, newline="") as fh:
        rd = csv.DictReader(fh)
        cols = rd.fieldnames
        for r in rd:
            rows.append(r)
print("trace columns:", cols)


def col(*names):
    for n in names:
        for c in cols:
            if c.lower() == n.lower():
                return c
    raise KeyError(names)


C_NAME, C_Q = col("Kernel_Name", "KernelName"), col("Queue_Id", "QueueId", "Queue_Index")
C_S, C_E = col("Start_Timestamp", "BeginNs", "Start"), col("End_Timestamp", "EndNs", "End")
C_STREAM = next((c for c in cols if c.lower() in ("stream_id", "streamid")), None)
rows.sort(key=lambda r: int(r[C_S]))


def short(n):
    n = n.split("(")[0]
    for k in ("fir_u8_direct", "fir_fmt_direct", "msk_demod", "blk_repair", "msg_split", "fill_random", "fir_u8_persist", "fir_u8_shared"):
        if k in n:
            return k
    return n[:40]


turns, cur = [], None
for r in rows:
    k = short(r[C_NAME])
    if k == "fill_random":
        if cur:
            turns.append(cur)
        cur = []
        continue
    if cur is not None:
        cur.append((k, r[C_Q], r.get(C_STREAM, "") if C_STREAM else "", int(r[C_S]), int(r[C_E])))
if cur:
    turns.append(cur)
turns = [t for t in turns if len(t) > 8]


def union(iv):
    iv = sorted(iv)
    out = []
    for s, e in iv:
        if out and s <= out[-1][1]:
            out[-1][1] = max(out[-1][1], e)
        else:
            out.append([s, e])
    return out


def inter_len(a, b):
    i = j = 0
    tot = 0
    while i < len(a) and j < len(b):
        s, e = max(a[i][0], b[j][0]), min(a[i][1], b[j][1])
        if e > s:
            tot += e - s
        if a[i][1] < b[j][1]:
            i += 1
        else:
            j += 1
    return tot


print("%d turns (contexts x rounds) found in %d dispatches" % (len(turns), len(rows)))
for ti, t in enumerate(turns):
    by = {}
    for k, q, st, s, e in t:
        by.setdefault(k, []).append((q, st, s, e))
    span = (max(x[4] for x in t) - min(x[3] for x in t)) / 1e6
    fir = union([(s, e) for k, q, st, s, e in t if k.startswith("fir")])
    msk = union([(s, e) for k, q, st, s, e in t if k == "msk_demod"])
    both = inter_len(fir, msk)
    busy = sum(e - s for s, e in union([(s, e) for k, q, st, s, e in t]))
    line = "turn %2d (context %d round %d): span %.1f ms, busy %.3f, fir busy %.3f, msk busy %.3f, both %.3f of span | " % (
        ti, ti % nctx, ti // nctx, span, busy / 1e6 / span, sum(e - s for s, e in fir) / 1e6 / span,
        sum(e - s for s, e in msk) / 1e6 / span, both / 1e6 / span)
    for k in sorted(by):
        v = by[k]
        line += "%s: n %d mean %.4f ms queues %s streams %s; " % (k, len(v), sum(e - s for _, _, s, e in v) / len(v) / 1e6,
                                                            sorted({q for q, _, _, _ in v}), sorted({st for _, st, _, _ in v}))
    print(line)
