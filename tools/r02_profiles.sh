# Round-2 evidence run: default bench (all configs + CPU baselines), kernel traces and PMC passes per config.
# Each profiler run sits under its own timeout; PMC passes carry --kernel-trace only.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02p
rm -rf $O; mkdir -p $O
cd $R
timeout 900 python bench.py > $O/bench_n1.json 2> $O/bench_n1.err; echo "bench rc=$?"
cd /tmp && export TMPDIR=/tmp
for c in c10_b512 sg2_32 sg2_512; do
  case $c in c10_b512) S="--steps 5 --warmup 3";; sg2_32) S="--steps 5 --warmup 3";; sg2_512) S="--steps 16 --warmup 2";; esac
  timeout 300 rocprofv3 --kernel-trace -d $O -o ${c}_kt -- python $R/bench.py --config $c $S --no-cpu-baseline > $O/${c}_kt.log 2>&1
  python $R/tools/rocpd_summary.py $O/${c}_kt_results.db > $O/${c}_kernel_trace.txt 2>&1
  grep -h '^{' $O/${c}_kt.log > $O/${c}_under_rocprof.json
  rm -f $O/${c}_kt_results.db
done
for c in ${PMC_CONFIGS-c10_b512 sg2_512}; do       # PMC_CONFIGS="" skips the counter passes
  case $c in c10_b512) S="--steps 3 --warmup 2";; sg2_512) S="--steps 2 --warmup 2";; esac
  P=$O/pmc_$c; mkdir -p $P
  B="python $R/bench.py --config $c $S --no-cpu-baseline --graph off"
  timeout 240 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $P -o fetch -- $B > $P/fetch.log 2>&1
  timeout 240 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $P -o write -- $B > $P/write.log 2>&1
  timeout 240 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F32 -d $P -o sq -- $B > $P/sq.log 2>&1
  timeout 240 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $P -o misc -- $B > $P/misc.log 2>&1
  cd $R && python tools/make_pmc_profile.py $P r02_${c}_n1 "bench.py --config $c $S --no-cpu-baseline --graph off" > $P/summary.txt 2>&1; cd /tmp
  rm -f $P/*.db
done
cd $R
for c in c10_b512 sg2_32 sg2_512; do cp $O/${c}_kernel_trace.txt profiles/r02_${c}_n1_kernel_trace.txt; cp $O/${c}_under_rocprof.json profiles/r02_${c}_n1_under_rocprof.json; done
tail -1 $O/bench_n1.json > profiles/r02_bench_n1.json
mkdir -p $O/profiles; cp profiles/r02_* $O/profiles/ 2>/dev/null
ls -la $O $O/profiles; head -c 1500 $O/bench_n1.json; echo; cat $O/pmc_c10_b512/summary.txt | head -30
