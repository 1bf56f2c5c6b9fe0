#!/bin/bash
# Run a list of GPU steps one after another on the GPU box, each under its own timeout, output to gpurun_out/<name>.log.
# An ordinary failure (a failing test) does not stop the chain; a step that TIMED OUT or was KILLED does (exit codes
# 124 / 137 / 143): after a hung GPU step no further GPU step is started.
#   tools/gpu_steps.sh  name1 seconds1 'command 1'  name2 seconds2 'command 2' ...
mkdir -p gpurun_out
export TMPDIR=/tmp
rc_all=0
while [ $# -ge 3 ]; do
  name=$1; secs=$2; cmd=$3; shift 3
  echo "=== step $name (limit ${secs}s): $cmd"
  start=$(date +%s)
  timeout -k 10 "$secs" bash -o pipefail -c "$cmd" > "gpurun_out/$name.log" 2>&1
  rc=$?
  echo "=== step $name rc=$rc after $(( $(date +%s) - start ))s"
  tail -n 6 "gpurun_out/$name.log"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ] || [ $rc -eq 143 ]; then
    echo "=== step $name was killed at its limit: stopping the chain"
    exit $rc
  fi
  [ $rc -ne 0 ] && rc_all=$rc
done
exit $rc_all
