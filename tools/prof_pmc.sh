# PMC passes for the roofline's `traffic` field (separate passes per MI355X_MICROARCH.md: FETCH_SIZE and WRITE_SIZE do
# not fit one pass; never combined with sys/hip/memory-copy traces).
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_bench
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O -o fetch -- python $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline > $O/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O -o write -- python $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline > $O/write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F32 -d $O -o sq -- python $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline > $O/sq.log 2>&1
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $O -o misc -- python $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline > $O/misc.log 2>&1
ls -la $O
