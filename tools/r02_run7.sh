R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02g
rm -rf $O; mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_graph_gpu.py tests/test_kernels_gpu.py tests/test_sndcgan_gpu.py -q -m gpu --timeout 600 -p no:cacheprovider > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
grep -n "^FAILED\|^ERROR\|passed\|failed\|Error" $O/pytest.log | head -20
for g in on off; do
  timeout 300 python bench.py --config c10_b512 --no-cpu-baseline --graph $g > $O/c10_$g.json 2> $O/c10_$g.err; echo "rc=$?"
  for b in 64 128; do timeout 300 python bench.py --config c10_b512 --no-cpu-baseline --graph $g --dev-local-batch $b > $O/c10_${g}_b$b.json 2> $O/c10_${g}_b$b.err; done
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r02g/c10_*.json')):
    try:
        d=json.load(open(f)); print(f.split('/')[-1], d['value'], d['ms_per_step'], d['config'].get('launch'), d['roofline']['frac'], d['roofline']['step_level']['frac'])
    except Exception as e: print(f, 'ERR', e, open(f.replace('.json','.err')).read()[-600:])
PY
