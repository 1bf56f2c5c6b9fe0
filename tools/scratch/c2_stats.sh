cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/c2stats -o c2 -- python3 $R/bench.py --steps 10 --warmup 3 --no-extras --no-side --cpu-pairs 0 --pin-schedule 0 > $R/gpurun_out/c2stats.log 2>&1
python3 $R/tools/prof_top.py $(find $R/gpurun_out/c2stats -name "*results.db" | head -1) 14 > $R/gpurun_out/c2stats.txt 2>&1
find $R/gpurun_out/c2stats -name '*.db' -delete
