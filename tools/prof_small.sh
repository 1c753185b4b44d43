# Kernel trace of one rank's share of an 8-GPU job (local batch 64) on a single GPU: how launch-bound is it?
mkdir -p gpurun_out/prof_small
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout 200 python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --dev-local-batch ${LB:-64} --force-dist --config c10_b512 --no-g-step 2>&1 | grep metric | cut -c1-260
timeout 240 rocprofv3 --kernel-trace -d $R/gpurun_out/prof_small -o s -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --dev-local-batch ${LB:-64} --force-dist --config c10_b512 --no-g-step > $R/gpurun_out/prof_small/bench.log 2>&1
grep metric $R/gpurun_out/prof_small/bench.log | cut -c1-260
python $R/tools/rocpd_summary.py $R/gpurun_out/prof_small/s_results.db > $R/gpurun_out/prof_small/kernel_trace.txt 2>&1; rm -f $R/gpurun_out/prof_small/s_results.db; head -45 $R/gpurun_out/prof_small/kernel_trace.txt | cut -c1-140
