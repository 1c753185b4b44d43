# PMC look at the HBM-bound kernels in isolation (tools/bench_hbm.py c10): where do the rgb convs wait?
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_small
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python $R/tools/bench_hbm.py c10"
timeout 120 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU -d $O -o sq -- $B > $O/sq.log 2>&1
timeout 120 rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE SQ_WAVES -d $O -o misc -- $B > $O/misc.log 2>&1
timeout 120 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O -o fetch -- $B > $O/fetch.log 2>&1
timeout 120 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O -o write -- $B > $O/write.log 2>&1
cd $R
for p in sq misc fetch write; do echo "== $p"; python tools/rocpd_pmc.py $O/${p}_results.db "" 2>&1 | grep -v "^at::" | head -80; done > $O/summary.txt
rm -f $O/*.db
cat $O/summary.txt | head -150
