#!/bin/bash
# dev: memory-system counters of the 32 -> 32 3x3 layer at 512^2 (48 images) next to the 128 -> 128 layer at 128^2 (same
# FLOPs): L1 <-> L2 request counts and latency, L2 hit/miss, L2 <-> fabric request sizes, L1 stall cycles, address translation
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc32m
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export CONV_ITERS=2 CONV_WARM=1 CONV_BATCH=48
export CONV_CUSTOM="512,32,32,3,1,1;128,128,128,3,1,1"
B="python $R/tools/bench_conv.py"
timeout 200 rocprofv3 --kernel-trace --pmc TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TOTAL_READ_sum TCP_TOTAL_WRITE_sum -d $O -o l1 -- $B > $O/l1.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum TCC_WRITE_sum TCC_TAG_STALL_sum -d $O -o l2 -- $B > $O/l2.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_RDREQ_DRAM_sum TCC_EA0_WRREQ_STALL_sum -d $O -o ea -- $B > $O/ea.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum -d $O -o st -- $B > $O/st.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_STALL_INFLIGHT_MAX_sum TA_TA_BUSY_sum TA_ADDR_STALL_BY_TC_CYCLES_sum -d $O -o tl -- $B > $O/tl.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_LEVEL_VMEM -d $O -o sq -- $B > $O/sq.log 2>&1
for k in l1 l2 ea st tl sq; do echo "== $k"; python $R/tools/rocpd_pmc.py $O/${k}_results.db igemm_lean 2>&1 | head -60; tail -2 $O/$k.log | cut -c1-200; done > $O/summary.txt 2>&1
rm -f $O/*.db
cat $O/summary.txt | head -230
