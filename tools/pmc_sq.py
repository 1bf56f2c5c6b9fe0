#!/usr/bin/env python3
"""Per-kernel SQ issue picture from a rocprofv3 --pmc results database (development tool).

    cd /tmp && rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY \
        SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU -d DIR -o pmc -- python3 tools/x.py
    python3 tools/pmc_sq.py DIR [kernel-name substring]

Prints, per kernel: launches, and every counter as a per-launch mean and as a fraction of SQ_WAVE_CYCLES."""
import collections
import glob
import os
import sqlite3
import sys

root = sys.argv[1]
want = sys.argv[2] if len(sys.argv) > 2 else ""
dbs = glob.glob(os.path.join(root, "**", "*results.db"), recursive=True)
if not dbs:
    raise SystemExit(f"no *results.db under {root}")
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for path in dbs:
    d = sqlite3.connect(path)
    cur = d.cursor()
    cols = [c[0] for c in cur.execute("select * from counters_collection limit 1").description]
    for r in cur.execute("select * from counters_collection"):
        row = dict(zip(cols, r))
        name = row["kernel_name"]
        if want in name:
            acc[name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][-90:]][row["counter_name"]].append(row["value"])
for k, cs in acc.items():
    n = max(len(v) for v in cs.values())
    wave = sum(cs.get("SQ_WAVE_CYCLES", [0])) / max(1, len(cs.get("SQ_WAVE_CYCLES", [0])))
    print(f"{k}  launches {n}")
    for c, v in sorted(cs.items()):
        mean = sum(v) / len(v)
        print(f"    {c:28s} {mean:16.0f}" + (f"   {mean / wave:6.3f} of wave cycles" if wave else ""))
