R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_async
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 240 rocprofv3 --kernel-trace -d $O -o a -- python $R/bench.py --steps 5 --warmup 3 --no-cpu-baseline > $O/bench.log 2>&1
grep -o '"ms_per_step": [0-9.]*' $O/bench.log | head -1
