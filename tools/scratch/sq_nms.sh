cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU -d $R/gpurun_out/sq1 -o pmc -- python3 $R/tools/kbench.py nms --images 448 --iters 3 > $R/gpurun_out/sq1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_BUSY_CYCLES -d $R/gpurun_out/sq2 -o pmc -- python3 $R/tools/kbench.py nms --images 448 --iters 3 > $R/gpurun_out/sq2.log 2>&1
python3 $R/tools/pmc_sq.py $R/gpurun_out/sq1 nms_ > $R/gpurun_out/sq1.txt
python3 $R/tools/pmc_sq.py $R/gpurun_out/sq2 nms_ > $R/gpurun_out/sq2.txt
find $R/gpurun_out/sq1 $R/gpurun_out/sq2 -name '*.db' -delete
