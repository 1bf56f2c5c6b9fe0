#!/bin/bash
# Same-box A/B of two builds of the library (development aid): bench with the tree's library, then with the
# libraries under .ab/old_lib copied over it (on the GPU box's scratch copy), alternating ROUNDS times.
#   gpurun -- 'bash tools/ab_bench.sh "--steps 300 --warmup 10 --cpu-pairs 0 --no-extras" 2'
# The comparison libraries come from another revision, built here on the CPU box first (.ab/ is git-ignored and travels
# with the snapshot):
#   mkdir -p .ab/src .ab/old_lib && git archive <rev> | tar -x -C .ab/src
#   (cd .ab/src && python -c "from onnx_image_processing_amd import build; build.build()")
#   cp .ab/src/onnx_image_processing_amd/lib/*.so .ab/old_lib/ && rm -rf .ab/src
set -e
ARGS=${1:---steps 300 --warmup 10 --cpu-pairs 0 --no-extras}
ROUNDS=${2:-2}
LIB=onnx_image_processing_amd/lib
mkdir -p gpurun_out/ab .ab/new_lib
cp $LIB/*.so .ab/new_lib/
for r in $(seq 1 $ROUNDS); do
  for v in new old; do
    cp .ab/${v}_lib/*.so $LIB/
    python bench.py $ARGS > gpurun_out/ab/${v}_$r.json 2> gpurun_out/ab/${v}_$r.err
    python tools/show_bench.py gpurun_out/ab/${v}_$r.json
  done
done
cp .ab/new_lib/*.so $LIB/
