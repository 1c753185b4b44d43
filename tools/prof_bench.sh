# Kernel trace of the default bench (every rocprofv3 run is wrapped in its own timeout: a failed profiler run can hang).
mkdir -p gpurun_out/prof3
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout 240 rocprofv3 --kernel-trace -d $R/gpurun_out/prof3 -o r3 -- python $R/bench.py --steps 5 --warmup 3 --no-cpu-baseline > $R/gpurun_out/prof3/bench.log 2>&1
ls $R/gpurun_out/prof3
