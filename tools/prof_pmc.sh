# PMC passes for the roofline's `traffic` field (separate passes per MI355X_MICROARCH.md: FETCH_SIZE and WRITE_SIZE do
# not fit one pass; never combined with sys/hip/memory-copy traces).  Every profiler run sits under its own timeout:
# a counter set the hardware cannot schedule makes rocprofv3 abort and then hang.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_bench
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline"
timeout 150 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O -o fetch -- $B > $O/fetch.log 2>&1
timeout 150 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O -o write -- $B > $O/write.log 2>&1
timeout 150 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F32 -d $O -o sq -- $B > $O/sq.log 2>&1
timeout 150 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $O -o misc -- $B > $O/misc.log 2>&1
ls -la $O
