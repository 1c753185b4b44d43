R=$GRAFT_REPO_ROOT
cd $R
timeout 1500 python -m pytest tests/test_stylegan2_gpu.py tests/test_stylegan2_512_gpu.py tests/test_stylegan2_gstep_gpu.py -q -m gpu --timeout 600 -p no:cacheprovider 2>&1 | grep -a "passed\|failed\|FAILED\|Error" | tail -5
for c in sg2_32 sg2_512; do timeout 300 python bench.py --config $c --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['config']['name'], d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['step_level']['frac'])"; done
