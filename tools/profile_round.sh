#!/bin/bash
# Round profile on the GPU box (run through gpurun from the repo root):
#   bench line, rocprofv3 kernel statistics of the same command, and the two PMC passes
#   (FETCH_SIZE, WRITE_SIZE -- separate runs, --kernel-trace only, as the guide prescribes).
# Outputs land in gpurun_out/round/; tools/profile_summarise.py turns them into profiles/<tag>_*.
set -e -o pipefail
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"
OUT="$ROOT/gpurun_out/round"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 python3 "$ROOT/bench.py" > "$OUT/bench.json" 2> "$OUT/bench.err"
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d "$OUT/stats" -o stats -- python3 "$ROOT/bench.py" --steps 5 --warmup 2 --cpu-pairs 0 > "$OUT/stats.log" 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d "$OUT/fetch" -o fetch -- python3 "$ROOT/bench.py" --steps 3 --warmup 1 --cpu-pairs 0 > "$OUT/fetch.log" 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d "$OUT/write" -o write -- python3 "$ROOT/bench.py" --steps 3 --warmup 1 --cpu-pairs 0 > "$OUT/write.log" 2>&1
echo profile_round done
