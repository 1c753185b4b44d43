R=$GRAFT_REPO_ROOT
cd $R
timeout 1500 python -m pytest tests/test_stylegan2_gpu.py tests/test_stylegan2_512_gpu.py tests/test_stylegan2_gstep_gpu.py -q -m gpu --timeout 600 -p no:cacheprovider 2>&1 | grep -a "passed\|failed\|FAILED" | tail -8
for i in 1 2; do timeout 300 python bench.py --config sg2_512 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['config']['name'], d['value'], d['ms_per_step'], d['roofline']['step_level']['frac'], d['config']['peak_hbm_gib'])"; done
