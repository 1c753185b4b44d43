# Round-2 GPU pass 1: full GPU test suite, default bench (all three configs), kernel traces of the StyleGAN2 configs.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02a
rm -rf $O; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -q -m gpu -x --timeout 600 > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -15 $O/pytest.log
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"
cat $O/bench_default.json | head -c 3000
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace -d $O -o sg2_512 -- python $R/bench.py --config sg2_512 --steps 16 --warmup 2 --no-cpu-baseline > $O/sg2_512.log 2>&1
timeout 200 rocprofv3 --kernel-trace -d $O -o sg2_32 -- python $R/bench.py --config sg2_32 --steps 10 --warmup 3 --no-cpu-baseline > $O/sg2_32.log 2>&1
cd $R
for c in sg2_512 sg2_32; do f=$(ls $O/${c}_results.db 2>/dev/null); [ -n "$f" ] && python tools/rocpd_summary.py $f > $O/${c}_kernel_trace.txt 2>&1; done
ls -la $O; head -30 $O/sg2_512_kernel_trace.txt
