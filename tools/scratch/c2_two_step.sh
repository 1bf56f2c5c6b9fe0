cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/c2ts -o c2 -- python3 $R/bench.py --two-step --steps 10 --warmup 3 --no-extras --no-side --cpu-pairs 0 --pin-schedule 0 > $R/gpurun_out/c2ts.log 2>&1
python3 $R/tools/prof_top.py $(find $R/gpurun_out/c2ts -name "*results.db" | head -1) 16 > $R/gpurun_out/c2ts.txt 2>&1
find $R/gpurun_out/c2ts -name '*.db' -delete
