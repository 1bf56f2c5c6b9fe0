cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R && rocprofv3 --kernel-trace --stats -d $R/gpurun_out/det -o d -- python3 $R/tools/scratch/detectors_profile.py > $R/gpurun_out/det.log 2>&1
python3 $R/tools/prof_top.py $(find $R/gpurun_out/det -name "*results.db" | head -1) 10 > $R/gpurun_out/det.txt 2>&1
find $R/gpurun_out/det -name '*.db' -delete
