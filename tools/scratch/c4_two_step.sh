cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
python3 $R/bench.py --workload c4 --pairs-per-gpu 128 --two-step --steps 10 --warmup 3 --no-extras --no-side --cpu-pairs 0 > $R/gpurun_out/c4ts.log 2>&1
python3 $R/bench.py --workload vo --pairs-per-gpu 128 --two-step --steps 10 --warmup 3 --no-extras --no-side --cpu-pairs 0 >> $R/gpurun_out/c4ts.log 2>&1
python3 $R/bench.py --workload c3 --pairs-per-gpu 128 --two-step --steps 10 --warmup 3 --no-extras --no-side --cpu-pairs 0 >> $R/gpurun_out/c4ts.log 2>&1
