R=$GRAFT_REPO_ROOT
cd $R
timeout 900 python -m pytest tests/test_igemm_gpu.py tests/test_stylegan2_gpu.py tests/test_stylegan2_512_gpu.py -q -m gpu --timeout 600 -p no:cacheprovider 2>&1 | grep -a "passed\|failed\|FAILED" | tail -5
echo "--- 256x32 on"; CONV_SET=sg2full CONV_BATCH=32 CONV_ITERS=5 CONV_WARM=2 CONV_LAYERS=0,1,2 timeout 200 python tools/bench_conv.py | head -3
echo "--- 256x32 off"; CONTRAD_IGEMM_NO256=1 CONV_SET=sg2full CONV_BATCH=32 CONV_ITERS=5 CONV_WARM=2 CONV_LAYERS=0,1,2 timeout 200 python tools/bench_conv.py | head -3
timeout 300 python bench.py --config sg2_512 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('sg2_512', d['value'], d['ms_per_step'], d['roofline']['step_level']['frac'])"
