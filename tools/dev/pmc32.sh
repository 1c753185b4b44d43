#!/bin/bash
# dev: counters of the 32 -> 32 3x3 layer at 512^2 (48 images) next to the 128 -> 128 layer at 128^2 (same FLOPs)
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc32
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export CONV_ITERS=2 CONV_WARM=1 CONV_BATCH=48
export CONV_CUSTOM="512,32,32,3,1,1;128,128,128,3,1,1"
timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU -d $O -o sq -- python $R/tools/bench_conv.py > $O/sq.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA -d $O -o in -- python $R/tools/bench_conv.py > $O/in.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC SQ_IFETCH SQ_INSTS_BRANCH -d $O -o mi -- python $R/tools/bench_conv.py > $O/mi.log 2>&1
for k in sq in mi; do echo "== $k"; python $R/tools/rocpd_pmc.py $O/${k}_results.db igemm_lean; done > $O/summary.txt 2>&1
rm -f $O/*.db
cat $O/summary.txt | head -150; tail -3 $O/mi.log
