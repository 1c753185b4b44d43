R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02i
rm -rf $O; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_igemm_gpu.py tests/test_kernels_gpu.py tests/test_sndcgan_gpu.py tests/test_stylegan2_gpu.py tests/test_graph_gpu.py -q -m gpu --timeout 600 -p no:cacheprovider 2>&1 | tail -3
for c in c10_b512 sg2_32 sg2_512; do timeout 300 python bench.py --config $c --no-cpu-baseline > $O/$c.json 2> $O/$c.err; done
python - <<'PY'
import json
for n in ('c10_b512','sg2_32','sg2_512'):
    try:
        d=json.load(open('gpurun_out/r02i/%s.json'%n)); print(n, d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['step_level']['frac'])
    except Exception as e: print(n, 'ERR', e)
PY
