#!/bin/bash
# Round profile on the GPU box (run through gpurun from the repo root):
#   bench line, rocprofv3 kernel statistics of the same command, the PMC passes the guide prescribes for HBM traffic
#   (FETCH_SIZE, WRITE_SIZE -- separate runs, --kernel-trace only) and the MFMA passes for the cost kernel
#   (busy cycles and F6F4 MOPS -- the packed-descriptor dot products run on the FP4 MFMA --, separate runs), then the same statistics for the one-pair-per-call path and for
#   BASELINE configs[2] / [3].  The program itself follows `--` (never a shell or env wrapper).
# Outputs land in gpurun_out/round/; tools/profile_summarise.py turns them into gpurun_out/round/summary/<tag>_*
# (copy those into profiles/).
set -e -o pipefail
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"
OUT="$ROOT/gpurun_out/round"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
B="$ROOT/bench.py"
timeout -k 10 600 python3 "$B" > "$OUT/bench.json" 2> "$OUT/bench.err"
echo bench done
# Every counted pass pins the Sinkhorn's stream schedule (--pin-schedule 0: the halves on {stream, helper}, what the tuner
# settles on without a communicator): with the tuner's nine trial calls in the counted window the per-launch averages
# of the row kernel mixed 224- and 448-pair launches (round 3's r03_bench_pmc_traffic.json: "157.6 MB per launch")
PIN="--pin-schedule 0 --no-side"
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d "$OUT/stats" -o stats -- python3 "$B" --steps 20 --warmup 3 --cpu-pairs 0 --no-extras $PIN > "$OUT/stats.log" 2>&1
echo stats done
for c in FETCH_SIZE WRITE_SIZE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F6F4; do
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $c -d "$OUT/pmc_$c" -o pmc -- python3 "$B" --steps 3 --warmup 1 --cpu-pairs 0 --no-extras $PIN > "$OUT/pmc_$c.log" 2>&1
  echo pmc $c done
done
# the same two traffic passes on uint8 frames (u8 ingest: K1 at 5 B/px, K4 on 1 KB windows)
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $c -d "$OUT/pmc_u8_$c" -o pmc -- python3 "$B" --frames u8 --steps 3 --warmup 1 --cpu-pairs 0 --no-extras $PIN > "$OUT/pmc_u8_$c.log" 2>&1
  echo pmc u8 $c done
done
# issue picture of every kernel: SQ counters in ONE pass (8 SQ slots)
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU -d "$OUT/pmc_SQ" -o pmc -- python3 "$B" --steps 3 --warmup 1 --cpu-pairs 0 --no-extras $PIN > "$OUT/pmc_SQ.log" 2>&1
echo pmc SQ done
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d "$OUT/latency" -o latency -- python3 "$ROOT/tools/latency_trace.py" --single-call --graph --iters 50 > "$OUT/latency.log" 2>&1
echo latency done
# BASELINE configs[2] (sparse pipeline and dense-BAD matcher) and configs[3]: bench line, kernel statistics and the two
# HBM-traffic passes each (so that their roofline objects carry a `traffic` figure)
for wl in c3 c3dense c4 vo; do
  timeout -k 10 300 python3 "$B" --workload $wl --pairs-per-gpu 128 > "$OUT/$wl.json" 2> "$OUT/$wl.err"
  timeout -k 10 300 rocprofv3 --kernel-trace --stats -d "$OUT/${wl}stats" -o $wl -- python3 "$B" --workload $wl --pairs-per-gpu 128 --steps 5 --warmup 2 --no-extras --cpu-pairs 0 --pin-schedule 0 > "$OUT/${wl}stats.log" 2>&1
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 300 rocprofv3 --kernel-trace --pmc $c -d "$OUT/pmc_${wl}_$c" -o pmc -- python3 "$B" --workload $wl --pairs-per-gpu 128 --steps 2 --warmup 1 --no-extras --cpu-pairs 0 --pin-schedule 0 > "$OUT/pmc_${wl}_$c.log" 2>&1
  done
  echo $wl done
done
# summarise here and drop the databases: gpurun copies at most 64 MiB back
python3 "$ROOT/tools/profile_summarise.py" "${PROFILE_TAG:-r04}" "$OUT/summary"
find "$OUT" -name '*.db' -delete
echo profile_round done
