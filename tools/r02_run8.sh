R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02h
rm -rf $O; mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_graph_gpu.py tests/test_kernels_gpu.py -q -m gpu --timeout 600 -p no:cacheprovider 2>&1 | tail -3
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace -d $O -o b64 -- python $R/bench.py --config c10_b512 --no-cpu-baseline --graph off --dev-local-batch 64 --steps 10 --warmup 3 > $O/b64.log 2>&1
cd $R
python tools/rocpd_summary.py $O/b64_results.db --timeline > $O/b64_kernel_trace.txt 2>&1
rm -f $O/*.db
head -50 $O/b64_kernel_trace.txt | cut -c1-140
