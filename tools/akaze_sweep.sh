#!/bin/bash
# On-box sweep of the fused AKAZE kernel's tile geometry (development tool; run through gpurun from the repo root):
#   rows per output tile (MI_AS_H) x waves per workgroup (MI_AS_WAVES); every variant is rebuilt and timed.
set -e
for cfg in "48 8" "40 8" "44 8" "52 8" "36 8" "48 6" "48 10" "32 8" "36 6" "40 10"; do
  set -- $cfg
  MI_BUILD_DEFINES="-DMI_AS_H=$1 -DMI_AS_WAVES=$2" python -m onnx_image_processing_amd.build > /dev/null 2>&1
  echo "H=$1 WAVES=$2: $(python tools/akaze_bench.py 128 20 2>&1 | grep 'scale 2' )"
done
python -m onnx_image_processing_amd.build --force > /dev/null 2>&1
