cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/vostats -o vo -- python3 $R/bench.py --workload vo --pairs-per-gpu 128 --steps 5 --warmup 2 --no-extras --cpu-pairs 0 --pin-schedule 0 > $R/gpurun_out/vostats.log 2>&1
python3 $R/tools/prof_top.py $(find $R/gpurun_out/vostats -name "*results.db" | head -1) 25 > $R/gpurun_out/vostats.txt 2>&1
find $R/gpurun_out/vostats -name '*.db' -delete
