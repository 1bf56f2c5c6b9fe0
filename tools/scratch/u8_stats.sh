cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/u8stats -o u8 -- python3 $R/bench.py --frames u8 --steps 10 --warmup 3 --no-extras --cpu-pairs 0 --pin-schedule 0 --no-side > $R/gpurun_out/u8stats.log 2>&1
python3 $R/tools/prof_top.py $(find $R/gpurun_out/u8stats -name "*results.db" | head -1) 12 > $R/gpurun_out/u8stats.txt 2>&1
find $R/gpurun_out/u8stats -name '*.db' -delete
