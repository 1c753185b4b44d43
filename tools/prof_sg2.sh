R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_sg2
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 280 rocprofv3 --kernel-trace -d $O -o sg2_512 -- python $R/tools/bench_sg2.py 512 > $O/sg2_512.log 2>&1
timeout 200 rocprofv3 --kernel-trace -d $O -o sg2_32 -- python $R/tools/bench_sg2.py 32 > $O/sg2_32.log 2>&1
tail -2 $O/sg2_512.log; tail -2 $O/sg2_32.log
