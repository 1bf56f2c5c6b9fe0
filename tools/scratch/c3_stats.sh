cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/c3stats -o c3 -- python3 $R/bench.py --workload c3 --pairs-per-gpu 128 --steps 5 --warmup 2 --no-extras --cpu-pairs 0 --pin-schedule 0 > $R/gpurun_out/c3stats.log 2>&1
python3 $R/tools/prof_top.py $(find $R/gpurun_out/c3stats -name "*results.db" | head -1) 14 > $R/gpurun_out/c3stats.txt 2>&1
find $R/gpurun_out/c3stats -name '*.db' -delete
