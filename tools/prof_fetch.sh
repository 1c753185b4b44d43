# One FETCH_SIZE pass over the default bench (HBM read traffic per kernel).
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_fetch
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 150 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O -o fetch -- python $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline > $O/fetch.log 2>&1
python $R/tools/rocpd_pmc.py $O/fetch_results.db igemm_lean
