R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02t
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace -d $O -o sg2_512 -- python $R/bench.py --config sg2_512 --steps 15 --warmup 2 --no-cpu-baseline > $O/sg2_512.log 2>&1
cd $R
python tools/rocpd_summary.py $O/sg2_512_results.db > $O/sg2_512_kernel_trace.txt 2>&1
rm -f $O/*.db
head -42 $O/sg2_512_kernel_trace.txt | cut -c1-130; grep TOTAL $O/sg2_512_kernel_trace.txt
