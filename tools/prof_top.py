#!/usr/bin/env python3
"""Print the per-kernel summary of a rocprofv3 rocpd database (development tool).

    rocprofv3 --kernel-trace --stats -d DIR -o NAME -- python3 <script>      # on the GPU box
    python tools/prof_top.py DIR/NAME_results.db [rows]
"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = int(sys.argv[2]) if len(sys.argv) > 2 else 20
print("# name, calls, total_ms, avg_us, pct")
for name, calls, total, avg, pct in db.execute("select * from top_kernels limit ?", (rows,)):
    print(f"{name[:110]}, {calls}, {total / 1e3:.3f}, {avg:.2f}, {pct:.2f}")
