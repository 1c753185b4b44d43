R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02e
rm -rf $O; mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_checkpoint_gpu.py -q -m gpu --timeout 600 -p no:cacheprovider > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
grep -n "^FAILED\|^ERROR\|passed\|failed" $O/pytest.log | head
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace -d $O -o sg2_512 -- python $R/bench.py --config sg2_512 --steps 16 --warmup 2 --no-cpu-baseline > $O/sg2_512.log 2>&1
timeout 200 rocprofv3 --kernel-trace -d $O -o sg2_32 -- python $R/bench.py --config sg2_32 --steps 10 --warmup 3 --no-cpu-baseline > $O/sg2_32.log 2>&1
cd $R
for c in sg2_512 sg2_32; do python tools/rocpd_summary.py $O/${c}_results.db --timeline > $O/${c}_kernel_trace.txt 2>&1; done
rm -f $O/*.db
head -40 $O/sg2_512_kernel_trace.txt | cut -c1-150
