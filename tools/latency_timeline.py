#!/usr/bin/env python3
"""Per-kernel durations and inter-kernel gaps of the last calls in a rocprofv3 kernel trace (development tool).

    python tools/latency_timeline.py DIR/lat_results.db [kernels_per_call]
"""
import sqlite3
import sys
from collections import defaultdict

db = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
view = [t for t in tabs if t == "kernels"] or [t for t in tabs if "kernel" in t.lower()]
rows = list(db.execute(f"select name, start, end from {view[0]} order by start"))
# calls are separated by host synchronisation: gaps > 30 us
calls, cur = [], []
for name, s, e in rows:
    if cur and s - cur[-1][2] > 30000:
        calls.append(cur)
        cur = []
    cur.append((name, s, e))
calls.append(cur)
calls = [c for c in calls if len(c) == len(calls[-2])][-10:]
print(f"{len(calls)} calls of {len(calls[-1])} kernels; span {sum(c[-1][2]-c[0][1] for c in calls)/len(calls)/1e3:.1f} us")
agg = defaultdict(lambda: [0, 0.0, 0.0])
for c in calls:
    prev = None
    for name, s, e in c:
        k = name.replace("void ", "").replace("(anonymous namespace)::", "")
        k = (k.split("(")[0] if "<" not in k.split("(")[0] else k[:k.index(">") + 1])[:70]
        agg[k][0] += 1
        agg[k][1] += (e - s) / 1e3
        if prev is not None:
            agg[k][2] += (s - prev) / 1e3
        prev = e
print("# kernel, launches/call, busy us/call, gap-before us/call")
for k, (n, busy, gap) in sorted(agg.items(), key=lambda kv: -(kv[1][1] + kv[1][2])):
    print(f"{k}, {n/len(calls):.1f}, {busy/len(calls):.1f}, {gap/len(calls):.1f}")
