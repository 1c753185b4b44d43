R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02u
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace -d $O -o sg2_32 -- python $R/bench.py --config sg2_32 --steps 10 --warmup 3 --no-cpu-baseline > $O/sg2_32.log 2>&1
cd $R
python tools/rocpd_summary.py $O/sg2_32_results.db --timeline > $O/sg2_32_kernel_trace.txt 2>&1
rm -f $O/*.db
head -36 $O/sg2_32_kernel_trace.txt | cut -c1-130; grep TOTAL $O/sg2_32_kernel_trace.txt
