R=$GRAFT_REPO_ROOT
cd $R
timeout 900 python -m pytest tests/test_graph_gpu.py tests/test_stylegan2_gpu.py -q -m gpu --timeout 600 -p no:cacheprovider 2>&1 | grep -a "passed\|failed\|FAILED\|Error\|error" | tail -12
for i in 1 2; do for c in sg2_32 sg2_512; do timeout 300 python bench.py --config $c --no-cpu-baseline 2>/tmp/err.txt | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['config']['name'], d['value'], d['ms_per_step'], d['config']['launch'], d['roofline']['step_level']['frac'])"; tail -2 /tmp/err.txt | cut -c1-300; done; done
