# Dev tool: hardware counters of the conv engine, one SNDCGAN layer per run (args: layer indices of tools/bench_conv.py).
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/conv_pmc
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export CONV_ITERS=2 CONV_WARM=1
for L in "$@"; do
  export CONV_LAYERS=$L
  timeout 120 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU -d $O -o sq$L -- python $R/tools/bench_conv.py > $O/sq$L.log 2>&1
  # (one memory counter per pass and a timeout on each: FETCH_SIZE + WRITE_SIZE + TCC_* + GRBM_GUI_ACTIVE in ONE pass aborted
  #  rocprofv3 and hung its finalizer until gpurun's limit, round 5)
  for cnt in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum" GRBM_GUI_ACTIVE; do
    timeout 120 rocprofv3 --kernel-trace --pmc $cnt -d $O -o mem${L}_$(echo $cnt | cut -d" " -f1) -- python $R/tools/bench_conv.py > $O/mem$L.log 2>&1
  done
  timeout 120 rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR -d $O -o lds$L -- python $R/tools/bench_conv.py > $O/lds$L.log 2>&1
  for k in sq mem${L}_FETCH_SIZE mem${L}_WRITE_SIZE mem${L}_TCC_HIT_sum mem${L}_GRBM_GUI_ACTIVE lds; do
    case $k in mem*) db=$O/${k}_results.db;; *) db=$O/${k}${L}_results.db;; esac
    echo "== layer $L $k"; [ -f $db ] && python $R/tools/rocpd_pmc.py $db igemm
  done
done > $O/summary.txt 2>&1
tail -5 $O/sq$1.log
