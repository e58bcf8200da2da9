#!/usr/bin/env python3
"""How much do the kernels of the engine calls in flight overlap?  Reads a rocprofv3 --kernel-trace CSV (GPU box:
    cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --output-format csv -d OUT -o t -- python3 bench.py --arch mnist --steps 4 --no-cpu-baseline --no-other-configs
    python tools/trace_overlap.py OUT/**/t_kernel_trace.csv)
and prints, over the last 60 % of the trace (the timed steps): the time at each concurrency level, and per kernel name its mean duration here."""
import csv, sys, collections, glob
f = sys.argv[1]
rows = list(csv.DictReader(open(f)))
ev = []
for r in rows:
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "")))
ev.sort()
t0, t1 = ev[0][0], max(e[1] for e in ev)
lo = t0 + int(0.4 * (t1 - t0))
ev = [e for e in ev if e[0] >= lo]
pts = []
for s, e, _ in ev:
    pts.append((s, 1)); pts.append((e, -1))
pts.sort()
lvl, last, hist = 0, pts[0][0], collections.Counter()
for t, d in pts:
    hist[lvl] += t - last
    last = t; lvl += d
tot = sum(hist.values())
print("window %.1f ms, %d dispatches" % (tot / 1e6, len(ev)))
for k in sorted(hist):
    print("  %d kernels running: %5.1f %%" % (k, 100.0 * hist[k] / tot))
dur = collections.defaultdict(list)
for s, e, n in ev:
    dur[n].append(e - s)
print("mean duration in this run (us), kernels above 1 % of the summed kernel time:")
allsum = sum(sum(v) for v in dur.values())
for n, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
    if sum(v) > 0.01 * allsum:
        print("  %-70s n=%5d mean %8.1f  share of summed kernel time %.3f" % (n[:70], len(v), sum(v) / len(v) / 1e3, sum(v) / allsum))
print("summed kernel time / window = %.2f (average number of kernels in flight)" % (allsum / tot))
# which queue do consecutive dispatches belong to?  (runs of one queue = the calls in flight are executed one after the other)
qcol = "Queue_Id" if "Queue_Id" in rows[0] else None
if qcol:
    evq = sorted((int(r["Start_Timestamp"]), r[qcol]) for r in rows if int(r["Start_Timestamp"]) >= lo)
    runs, cur, n = [], None, 0
    for _, q in evq:
        if q == cur: n += 1
        else:
            if cur is not None: runs.append((cur, n))
            cur, n = q, 1
    runs.append((cur, n))
    print("queues seen:", sorted({q for _, q in evq}), " dispatches per run of one queue: first 40 runs", [n for _, n in runs[:40]])
    print("columns:", list(rows[0].keys()))
