R=$GRAFT_REPO_ROOT
cd $R
timeout 900 python -m pytest tests/test_igemm_gpu.py tests/test_sndcgan_gpu.py tests/test_stylegan2_gpu.py tests/test_stylegan2_512_gpu.py -q -m gpu --timeout 600 -p no:cacheprovider 2>&1 | grep -a "passed\|failed\|FAILED" | tail -5
for g in 8 16 32 64; do echo "cgroup $g"; CONTRAD_DGRAD_CGROUP=$g CONV_CUSTOM="257,64,128,3,2,0;65,256,512,3,2,0;129,128,256,3,2,0;33,512,512,3,2,0;32,64,128,4,2,1;16,128,256,4,2,1" CONV_BATCH=32 CONV_ITERS=5 CONV_WARM=2 timeout 100 python tools/bench_conv.py | grep GF | cut -c50-95; done
for c in c10_b512 sg2_32 sg2_512; do timeout 300 python bench.py --config $c --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['config']['name'], d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['step_level']['frac'])"; done
