#!/usr/bin/env python3
"""Per-kernel table of the second half (= the hipGraph replays) of a `rocprofv3 --kernel-trace` run, from its rocpd database:
    python tools/kernel_table.py <results.db>
kernel name, launches, average duration, share of the busy time; the busy fraction of the span and the median gap between consecutive kernels."""
import collections, sqlite3, statistics, sys

db = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
disp = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
sym = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
names = {r[0]: r[1] for r in db.execute(f"select id, kernel_name from {sym}")}
rows = list(db.execute(f"select kernel_id, start, end from {disp} order by start"))
rows = rows[len(rows) // 2:]
busy, span = sum(e - s for _, s, e in rows), rows[-1][2] - rows[0][1]
gaps = [rows[i + 1][1] - rows[i][2] for i in range(len(rows) - 1)]
print(f"{len(rows)} launches, busy {busy / span:.3f} of the span, median gap {statistics.median(gaps)} ns, mean duration {busy / len(rows) / 1e3:.1f} us")
acc = collections.defaultdict(lambda: [0, 0])
for k, s, e in rows:
    acc[names[k]][0] += 1; acc[names[k]][1] += e - s
for n, (c, t) in sorted(acc.items(), key=lambda x: -x[1][1])[:40]:
    print(f"{n[:86]:86s} {c:6d} {t / c / 1e3:8.1f} us {100 * t / busy:5.1f}%")
