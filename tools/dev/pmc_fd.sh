#!/bin/bash
# dev: why is the pixel-major FWD of an 8x8 map slower than its DGRAD (same GEMM shape, same skipped taps)?  Counters of both
# (kernel names differ in the MODE template argument).  CONV_CUSTOM / CONV_BATCH may be overridden.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_fd
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export CONV_ITERS=2 CONV_WARM=1 CONV_BATCH=${CONV_BATCH:-1536} CONV_MODES=fwd,dgrad
export CONV_CUSTOM=${CONV_CUSTOM:-"8,256,256,3,1,1"}
B="python $R/tools/bench_conv.py"
timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU -d $O -o sq -- $B > $O/sq.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA -d $O -o in -- $B > $O/in.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_BRANCH -d $O -o mi -- $B > $O/mi.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum -d $O -o tc -- $B > $O/tc.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE WRITE_SIZE TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum -d $O -o hb -- $B > $O/hb.log 2>&1
for k in sq in mi tc hb; do echo "== $k"; python $R/tools/rocpd_pmc.py $O/${k}_results.db igemm_lean; done > $O/summary.txt 2>&1
rm -f $O/*.db
cat $O/summary.txt | head -150; tail -3 $O/hb.log
