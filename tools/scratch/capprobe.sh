#!/bin/bash
for c in "hipEventQuery relaxed" "hipEventElapsedTime relaxed" "hipStreamWaitEvent" "hipGetLastError" "lib: 64 pairs, NO_FORK" "lib: 64 pairs, pinned unsplit" "lib: 64 pairs, pinned fork" "lib: 64 pairs, tuning" "hipStreamSynchronize"; do
  timeout -k 5 60 python tools/capture_mode_probe.py "$c" 2>&1 | grep -E "SURVIVED|INVALIDATED" | cut -c1-200
done
