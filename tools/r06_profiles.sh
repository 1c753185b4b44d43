# Round-6 evidence run (on the GPU box, via gpurun): per config a
# rocprofv3 kernel trace of `bench.py --config <c> --no-g-step` (D-step launches only: the generator-step side
# measurement would mix other shapes into the same kernel names) joined with the bench's own shape table
# (tools/rocpd_rows.py -> one row per (kernel, layer shape) with GFLOP per launch), and the PMC passes (own runs,
# --kernel-trace only beside the counters), then the default bench (all configs + CPU baselines).  Usage: PMC_CONFIGS="c10_b512 sg2_32 sg2_512" bash tools/r06_profiles.sh [tag]
R=$GRAFT_REPO_ROOT
TAG=${1:-r06}
O=$R/gpurun_out/${TAG}p
rm -rf $O; mkdir -p $O
cd $R
if [ -z "$SKIP_TESTS" ]; then     # the whole GPU suite first, every tolerance with the error actually observed (tests/conftest.py: margin)
  (cd $R; rm -f $O/margins.txt; CONTRAD_MARGINS=$O/margins.txt timeout 1200 python -m pytest tests -q -m gpu > $O/pytest.log 2>&1; tail -3 $O/pytest.log)
fi
cd /tmp && export TMPDIR=/tmp
for c in ${KT_CONFIGS-c10_b512 sg2_32 sg2_512}; do
  case $c in c10_b512) S="--steps 5 --warmup 3";; sg2_32) S="--steps 5 --warmup 3";; sg2_512) S="--steps 16 --warmup 2";; esac
  timeout 400 rocprofv3 --kernel-trace -d $O -o ${c}_kt -- python $R/bench.py --config $c $S --no-cpu-baseline --no-g-step --shape-table $O/${c}_shapes.json > $O/${c}_kt.log 2>&1
  python $R/tools/rocpd_summary.py $O/${c}_kt_results.db > $O/${c}_kernel_trace.txt 2>&1
  python $R/tools/rocpd_rows.py $O/${c}_kt_results.db $O/${c}_shapes.json > $O/${c}_rows.txt 2>&1
  grep -h '^{' $O/${c}_kt.log > $O/${c}_under_rocprof.json
  rm -f $O/${c}_kt_results.db
done
for c in ${PMC_CONFIGS-c10_b512 sg2_32 sg2_512}; do       # PMC_CONFIGS="" skips the counter passes
  case $c in c10_b512) S="--steps 3 --warmup 2";; sg2_32) S="--steps 3 --warmup 2";; sg2_512) S="--steps 2 --warmup 2";; esac
  P=$O/pmc_$c; mkdir -p $P
  B="python $R/bench.py --config $c $S --no-cpu-baseline --no-g-step --graph off"
  timeout 240 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $P -o fetch -- $B > $P/fetch.log 2>&1
  timeout 240 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $P -o write -- $B > $P/write.log 2>&1
  timeout 240 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F32 -d $P -o sq -- $B > $P/sq.log 2>&1
  timeout 240 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $P -o misc -- $B > $P/misc.log 2>&1
  timeout 240 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum -d $P -o tcc -- $B > $P/tcc.log 2>&1     # L2 hit rate (round 4)
  cd $R && python tools/make_pmc_profile.py $P ${TAG}_${c}_n1 "bench.py --config $c $S --no-cpu-baseline --no-g-step --graph off" > $P/summary.txt 2>&1; cd /tmp
  rm -f $P/*.db
done
# one rank of the 2- / 4- / 8-GPU headline config on one GPU (per-rank batch 256 / 128 / 64; every collective on a 1-rank RCCL
# group): kernel trace + per-shape rows, then the un-profiled step time
for B in ${RANK_BATCHES-256 128 64}; do
  timeout 300 rocprofv3 --kernel-trace -d $O -o b${B}_kt -- python $R/bench.py --config c10_b512 --dev-local-batch $B --force-dist --steps 20 --warmup 3 --no-cpu-baseline --no-g-step --shape-table $O/b${B}_shapes.json > $O/b${B}_kt.log 2>&1
  python $R/tools/rocpd_summary.py $O/b${B}_kt_results.db > $O/b${B}_kernel_trace.txt 2>&1
  python $R/tools/rocpd_rows.py $O/b${B}_kt_results.db $O/b${B}_shapes.json > $O/b${B}_rows.txt 2>&1
  rm -f $O/b${B}_kt_results.db
  (cd $R; timeout 200 python bench.py --config c10_b512 --dev-local-batch $B --force-dist --steps 200 --warmup 5 --no-cpu-baseline --no-g-step 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('# per-rank batch $B on one GPU (200 replayed steps, not under the profiler): %.3f ms per step, launch: %s' % (d['ms_per_step'], d['config']['launch']))") >> $O/b${B}_rows.txt
done
cd $R
if [ -z "$SKIP_BENCH" ]; then     # after the counter passes: its roofline.traffic then comes from THIS run's PMC summaries
  timeout 900 python bench.py > $O/bench_n1.json 2> $O/bench_n1.err; echo "bench rc=$?"
fi
mkdir -p $O/profiles
for c in ${KT_CONFIGS-c10_b512 sg2_32 sg2_512}; do
  cp $O/${c}_kernel_trace.txt $O/profiles/${TAG}_${c}_n1_kernel_trace.txt
  cp $O/${c}_rows.txt $O/profiles/${TAG}_${c}_n1_rows.txt
  cp $O/${c}_shapes.json $O/profiles/${TAG}_${c}_n1_shapes.json
  cp $O/${c}_under_rocprof.json $O/profiles/${TAG}_${c}_n1_under_rocprof.json
done
[ -f $O/bench_n1.json ] && tail -1 $O/bench_n1.json > $O/profiles/${TAG}_bench_n1.json
[ -f $O/margins.txt ] && cp $O/margins.txt $O/profiles/${TAG}_test_margins.txt
[ -f $O/pytest.log ] && grep -E " passed| failed| error" $O/pytest.log | tail -3 > $O/profiles/${TAG}_pytest_tail.txt
cp profiles/${TAG}_*_pmc.* $O/profiles/ 2>/dev/null
for B in ${RANK_BATCHES-256 128 64}; do
  cp $O/b${B}_rows.txt $O/profiles/${TAG}_c10_b${B}_rank_rows.txt; cp $O/b${B}_kernel_trace.txt $O/profiles/${TAG}_c10_b${B}_rank_kernel_trace.txt; cp $O/b${B}_shapes.json $O/profiles/${TAG}_c10_b${B}_rank_shapes.json
done
ls -la $O/profiles; [ -f $O/bench_n1.json ] && head -c 600 $O/bench_n1.json; echo
for c in ${KT_CONFIGS-c10_b512 sg2_32 sg2_512}; do head -30 $O/${c}_rows.txt; done
