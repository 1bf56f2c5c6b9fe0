cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R && rocprofv3 --kernel-trace --stats -d $R/gpurun_out/dctor -o d -- python3 $R/tools/scratch/default_ctor_profile.py > $R/gpurun_out/dctor.log 2>&1
python3 $R/tools/prof_top.py $(find $R/gpurun_out/dctor -name "*results.db" | head -1) 22 > $R/gpurun_out/dctor.txt 2>&1
find $R/gpurun_out/dctor -name '*.db' -delete
