R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02d
rm -rf $O; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_stylegan2_gpu.py tests/test_stylegan2_512_gpu.py tests/test_stylegan2_gstep_gpu.py tests/test_checkpoint_gpu.py tests/test_integration_gpu.py -q -m gpu --timeout 600 -p no:cacheprovider > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
grep -n "^FAILED\|^ERROR\|passed\|failed" $O/pytest.log | head -40
timeout 300 python bench.py --config sg2_512 --no-cpu-baseline > $O/sg2_512.json 2> $O/sg2_512.err; echo "rc=$?"
timeout 300 python bench.py --config sg2_32 --no-cpu-baseline > $O/sg2_32.json 2> $O/sg2_32.err; echo "rc=$?"
python - <<'PY'
import json
for n in ('sg2_512','sg2_32'):
    try:
        d=json.load(open('gpurun_out/r02d/%s.json'%n)); print(n, d['value'], d['ms_per_step'], d['roofline']['step_level'])
    except Exception as e: print(n, 'ERR', e)
PY
