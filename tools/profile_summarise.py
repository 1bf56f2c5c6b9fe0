#!/usr/bin/env python3
"""Turn gpurun_out/round/ (tools/profile_round.sh) into the committed summaries under profiles/.

    python tools/profile_summarise.py r01        # writes profiles/r01_bench.json, _kernel_stats.csv, _pmc_traffic.json
"""
import collections
import json
import os
import sqlite3
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", "round")
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
dst = os.path.join(ROOT, "profiles")


def short(name):
    """kernel name without namespaces and argument list (template arguments kept)"""
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    depth, out = 0, []
    for ch in name:
        if ch == "<":
            depth += 1
        elif ch == ">":
            depth -= 1
        elif ch == "(" and depth == 0:
            break
        out.append(ch)
    return "".join(out).replace(", ", ",")


line = open(os.path.join(SRC, "bench.json")).read().strip().splitlines()[-1]
PAIRS = int(json.loads(line)["config"]["pairs_per_gpu_per_step"])
open(os.path.join(dst, f"{tag}_bench.json"), "w").write(line + "\n")

db = sqlite3.connect(os.path.join(SRC, "stats", "stats_results.db"))
with open(os.path.join(dst, f"{tag}_bench_kernel_stats.csv"), "w") as f:
    f.write("# rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 5 --warmup 2 --cpu-pairs 0   "
            f"(1 x MI355X, {PAIRS} pairs/step)\n# name, calls, total_ms, avg_us, pct\n")
    for name, calls, total, avg, pct in db.execute("select * from top_kernels"):
        f.write(f"{short(name)}, {calls}, {total / 1e3:.3f}, {avg:.2f}, {pct:.2f}\n")

traffic = collections.defaultdict(dict)
for which, counter in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
    d = sqlite3.connect(os.path.join(SRC, which, f"{which}_results.db"))
    cur = d.cursor()
    cols = [c[0] for c in cur.execute("select * from counters_collection limit 1").description]
    acc = collections.defaultdict(list)
    for r in cur.execute("select * from counters_collection"):
        row = dict(zip(cols, r))
        if row["counter_name"] == counter:
            acc[short(row["kernel_name"])].append(row["value"])
    for k, v in acc.items():
        traffic[k][counter + "_KB_raw_per_launch"] = sum(v) / len(v)
        traffic[k]["launches"] = len(v)
out = {"note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over `bench.py --steps 3 --warmup 1` "
               f"({PAIRS} pairs = 2 x {PAIRS} images per step); per-launch averages; counters are KB; reads doubled per the gfx950 "
               "note in MI355X_MICROARCH.md (FETCH_SIZE reports half of wide coalesced reads)",
       "pairs_per_gpu": PAIRS, "kernels": {}}
for k, v in sorted(traffic.items(), key=lambda kv: -kv[1].get("FETCH_SIZE_KB_raw_per_launch", 0)):
    if k.startswith("at::") or k.startswith("__amd") or "elementwise" in k or "Cat" in k or "reduce_kernel" in k:
        continue
    rd = v.get("FETCH_SIZE_KB_raw_per_launch", 0.0) * 1024 * 2 / 1e6
    wr = v.get("WRITE_SIZE_KB_raw_per_launch", 0.0) * 1024 / 1e6
    out["kernels"][k] = {**{kk: round(vv, 3) for kk, vv in v.items()}, "read_MB_corrected_x2": round(rd, 1),
                         "write_MB": round(wr, 1), "total_MB": round(rd + wr, 1)}
json.dump(out, open(os.path.join(dst, f"{tag}_bench_pmc_traffic.json"), "w"), indent=1)
print("wrote", tag)
