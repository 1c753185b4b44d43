# Round-2 GPU pass 2: full GPU test suite (no -x), default bench (all three configs).
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02b
rm -rf $O; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -q -m gpu --timeout 600 -p no:cacheprovider > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -40 $O/pytest.log | cut -c1-250
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"
tail -3 $O/bench_default.err
