#!/usr/bin/env python3
"""Turn gpurun_out/round/ (tools/profile_round.sh) into the committed summaries under profiles/.

    python tools/profile_summarise.py r01        # writes profiles/r01_bench.json, _kernel_stats.csv, _pmc_traffic.json
    python tools/profile_summarise.py r01 DIR    # ... into DIR (profile_round.sh summarises on the GPU box, because the
                                                 # rocprofv3 databases are too large to travel back)
"""
import collections
import json
import os
import sqlite3
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", "round")
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
dst = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "profiles")
os.makedirs(dst, exist_ok=True)


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

def stats_csv(sub, base, out_name, header):
    path = os.path.join(SRC, sub, f"{base}_results.db")
    if not os.path.exists(path):
        return
    db = sqlite3.connect(path)
    with open(os.path.join(dst, out_name), "w") as f:
        f.write(header + "\n# name, calls, total_ms, avg_us, pct\n")
        for name, calls, total, avg, pct in db.execute("select * from top_kernels"):
            f.write(f"{short(name)}, {calls}, {total / 1e3:.3f}, {avg:.2f}, {pct:.2f}\n")


stats_csv("stats", "stats", f"{tag}_bench_kernel_stats.csv",
          "# rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 20 --warmup 3 --cpu-pairs 0 --no-extras --pin-schedule 0 --no-side   "
          f"(1 x MI355X, {PAIRS} pairs/step)")
stats_csv("latency", "latency", f"{tag}_latency_kernel_stats.csv",
          "# rocprofv3 --kernel-trace --stats -- python3 tools/latency_trace.py --single-call --graph --iters 50   "
          "(ONE 640x480 K=512 pair per call through mi_match_pairs, replayed as a hipGraph; every call synchronised)")
def side_traffic(wl, pairs):
    """FETCH_SIZE / WRITE_SIZE passes of a side workload -> {tag}_{wl}_pmc_traffic.json (same corrections as the main one)"""
    t = collections.defaultdict(dict)
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        path = os.path.join(SRC, f"pmc_{wl}_{counter}", "pmc_results.db")
        if not os.path.exists(path):
            return
        cur = sqlite3.connect(path).cursor()
        cols = [c[0] for c in cur.execute("select * from counters_collection limit 1").description]
        acc = collections.defaultdict(list)
        for r in cur.execute("select * from counters_collection"):
            row = dict(zip(cols, r))
            if row["counter_name"] == counter:
                acc[short(row["kernel_name"])].append(row["value"])
        for k, v in acc.items():
            t[k][counter + "_KB_raw_per_launch"] = sum(v) / len(v)
            t[k]["launches"] = len(v)
    o = {"note": f"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes over `bench.py --workload {wl} --pairs-per-gpu {pairs} "
                 "--steps 2 --warmup 1`; per-launch averages; counters are KB; reads doubled per the gfx950 note in MI355X_MICROARCH.md",
         "workload": wl, "pairs_per_gpu": pairs, "kernels": {}}
    for k, v in sorted(t.items(), key=lambda kv: -kv[1].get("FETCH_SIZE_KB_raw_per_launch", 0)):
        if k.startswith("at::") or k.startswith("__amd") or "elementwise" in k or "Cat" in k or "reduce_kernel" in k:
            continue
        rd = v.get("FETCH_SIZE_KB_raw_per_launch", 0.0) * 1024 * 2 / 1e6
        wr = v.get("WRITE_SIZE_KB_raw_per_launch", 0.0) * 1024 / 1e6
        o["kernels"][k] = {**{kk: round(vv, 3) for kk, vv in v.items()}, "read_MB_corrected_x2": round(rd, 1),
                           "write_MB": round(wr, 1), "total_MB": round(rd + wr, 1)}
    json.dump(o, open(os.path.join(dst, f"{tag}_{wl}_pmc_traffic.json"), "w"), indent=1)


for wl, pairs in (("c3", 128), ("c3dense", 128), ("c4", 128), ("vo", 128)):
    side_traffic(wl, pairs)
    stats_csv(wl + "stats", wl, f"{tag}_{wl}_kernel_stats.csv",
              f"# rocprofv3 --kernel-trace --stats -- python3 bench.py --workload {wl} --pairs-per-gpu {pairs} --steps 5 --warmup 2 --no-extras --cpu-pairs 0 --pin-schedule 0")
    src = os.path.join(SRC, wl + ".json")
    if os.path.exists(src):
        open(os.path.join(dst, f"{tag}_{wl}_bench.json"), "w").write(open(src).read().strip().splitlines()[-1] + "\n")

traffic = collections.defaultdict(dict)
def pmc_rows(counter):
    d = sqlite3.connect(os.path.join(SRC, "pmc_" + counter, "pmc_results.db"))
    cur = d.cursor()
    cols = [c[0] for c in cur.execute("select * from counters_collection limit 1").description]
    acc = collections.defaultdict(list)
    for r in cur.execute("select * from counters_collection"):
        row = dict(zip(cols, r))
        if row["counter_name"] == counter:
            acc[short(row["kernel_name"])].append(row["value"])
    return acc


for which, counter, sub in (("fetch", "FETCH_SIZE", "pmc_FETCH_SIZE"), ("write", "WRITE_SIZE", "pmc_WRITE_SIZE"),
                            ("fetch", "FETCH_SIZE", "pmc_u8_FETCH_SIZE"), ("write", "WRITE_SIZE", "pmc_u8_WRITE_SIZE")):
    if not os.path.exists(os.path.join(SRC, sub, "pmc_results.db")):
        continue                                                 # the uint8 passes are optional
    d = sqlite3.connect(os.path.join(SRC, sub, "pmc_results.db"))
    cur = d.cursor()
    cols = [c[0] for c in cur.execute("select * from counters_collection limit 1").description]
    acc = collections.defaultdict(list)
    for r in cur.execute("select * from counters_collection"):
        row = dict(zip(cols, r))
        if row["counter_name"] == counter:
            acc[short(row["kernel_name"])].append(row["value"])
    for k, v in acc.items():
        if sub.startswith("pmc_u8_") and counter + "_KB_raw_per_launch" in traffic.get(k, {}):
            continue                                             # kernels that do not depend on the pixel type: keep the fp32 run's
        traffic[k][counter + "_KB_raw_per_launch"] = sum(v) / len(v)
        traffic[k]["launches"] = len(v)
out = {"note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over `bench.py --steps 3 --warmup 1 --pin-schedule 0` (the Sinkhorn stream schedule pinned: every row-kernel launch is a half batch; and over "
               "`--frames u8` for the kernels that read uint8 pixels) "
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

# MFMA counters of the cost kernel (north_star: "MFMA utilisation on the cost matrix"); three separate passes
try:
    busy, cu, mops = pmc_rows("SQ_VALU_MFMA_BUSY_CYCLES"), pmc_rows("SQ_BUSY_CU_CYCLES"), pmc_rows("SQ_INSTS_VALU_MFMA_MOPS_F6F4")
    mf = {"note": "rocprofv3 --pmc, one counter per pass over `bench.py --steps 3 --warmup 1 --no-extras --pin-schedule 0`; per-launch sums over the "
                  "chip.  SQ_VALU_MFMA_BUSY_CYCLES / (SQ_BUSY_CU_CYCLES * 4 SIMDs) is the fraction of the kernel's busy "
                  "SIMD-cycles with the MFMA pipe occupied; MOPS_F6F4 * 512 = multiply-add operations issued on the FP4 MFMA (the packed "
                  "descriptors' dot products: a bit = the nibble 1.0 / 0.0, exact fp32 sums)",
          "pairs_per_gpu": PAIRS, "kernels": {}}
    for k in busy:
        if "cost" not in k:
            continue
        b = sum(busy[k]) / len(busy[k])
        c = sum(cu.get(k, [0])) / max(1, len(cu.get(k, [0])))
        m = sum(mops.get(k, [0])) / max(1, len(mops.get(k, [0])))
        mf["kernels"][k] = {"SQ_VALU_MFMA_BUSY_CYCLES": b, "SQ_BUSY_CU_CYCLES": c, "SQ_INSTS_VALU_MFMA_MOPS_F6F4": m,
                            "mfma_busy_fraction_of_busy_simd_cycles": (b / (4.0 * c)) if c else None,
                            "fp4_ops_issued": m * 512.0, "ops_algorithmic": 2.0 * PAIRS * 512 ** 3}
    json.dump(mf, open(os.path.join(dst, f"{tag}_bench_pmc_mfma.json"), "w"), indent=1)
except Exception as e:      # noqa: BLE001
    print("mfma summary skipped:", e)
print("wrote", tag)


# SQ issue counters of every kernel of the batched step (one pass): fractions of the kernel's wave cycles
try:
    d = sqlite3.connect(os.path.join(SRC, "pmc_SQ", "pmc_results.db"))
    cur = d.cursor()
    cols = [c[0] for c in cur.execute("select * from counters_collection limit 1").description]
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in cur.execute("select * from counters_collection"):
        row = dict(zip(cols, r))
        k = short(row["kernel_name"])
        if k.startswith("at::") or k.startswith("__amd") or "elementwise" in k or "Cat" in k or "reduce_kernel" in k:
            continue
        acc[k][row["counter_name"]].append(row["value"])
    sq = {"note": "rocprofv3 --pmc, eight SQ counters in one pass over `bench.py --steps 3 --warmup 1 --no-extras --pin-schedule 0`; per-launch "
                  "sums over the chip (SQ_WAVE_CYCLES, SQ_WAIT_*, SQ_ACTIVE_INST_* count quad-cycles per wave) and their "
                  "fraction of the kernel's wave cycles: WAIT_ANY = parked on s_waitcnt / barrier, WAIT_INST_ANY = ready but "
                  "not issued, ACTIVE_INST_ANY = issuing; LDS_BANK_CONFLICT / LDS_IDX_ACTIVE = conflict share of the LDS array's "
                  "busy cycles", "pairs_per_gpu": PAIRS, "kernels": {}}
    for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1].get("SQ_WAVE_CYCLES", [0]))):
        wc = sum(v["SQ_WAVE_CYCLES"]) / len(v["SQ_WAVE_CYCLES"])
        ent = {"launches": len(v["SQ_WAVE_CYCLES"])}
        for c, vals in sorted(v.items()):
            m = sum(vals) / len(vals)
            ent[c] = round(m)
            ent[c + "_per_wave_cycle"] = round(m / wc, 4) if wc else None
        sq["kernels"][k] = ent
    json.dump(sq, open(os.path.join(dst, f"{tag}_bench_pmc_sq.json"), "w"), indent=1)
except Exception as e:                                           # the pass is optional
    print("no SQ pass:", e)
