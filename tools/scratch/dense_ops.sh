cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R && rocprofv3 --kernel-trace --stats -d $R/gpurun_out/dense -o d -- python3 $R/tools/scratch/dense_ops_profile.py > $R/gpurun_out/dense.log 2>&1
python3 $R/tools/prof_top.py $(find $R/gpurun_out/dense -name "*results.db" | head -1) 16 > $R/gpurun_out/dense.txt 2>&1
find $R/gpurun_out/dense -name '*.db' -delete
