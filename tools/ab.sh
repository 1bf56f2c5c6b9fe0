#!/bin/bash
# A/B timing of two builds of the library on ONE GPU box (boxes differ by a few percent):
#   tools/ab.sh <base.so> "<command>"    runs <command> alternately with MI355X_MATCH_LIB=<base.so> and the in-tree build
# Output under gpurun_out/ab_{base,new}_{1,2}.log
set -e
BASE=$1; shift
for round in 1 2; do
  MI355X_MATCH_LIB=$BASE bash -c "$*" > gpurun_out/ab_base_$round.log 2>&1
  bash -c "$*" > gpurun_out/ab_new_$round.log 2>&1
done
for f in gpurun_out/ab_base_1.log gpurun_out/ab_new_1.log gpurun_out/ab_base_2.log gpurun_out/ab_new_2.log; do echo "== $f"; tail -${AB_TAIL:-6} $f | cut -c1-220; done
