#!/bin/bash
# dev: stride-1 DGRAD split-K -- parity, per-layer rates at 3N = 192, the per-rank step at batch 64
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_igemm_gpu.py -x -q 2>&1 | tail -5
export CONV_ITERS=30 CONV_WARM=5 CONV_BATCH=192
export CONV_CUSTOM="4,512,512,3,1,1;8,256,256,3,1,1;16,128,128,3,1,1;1,8192,1536,1,1,0;4,528,512,3,1,1"
for v in 1 0; do echo "== CONTRAD_IGEMM_SPLITK=$v"; CONTRAD_IGEMM_SPLITK=$v timeout 120 python tools/bench_conv.py 2>&1 | grep "^H"; done
for v in 1 0; do echo "== step at B=64, SPLITK=$v"; CONTRAD_IGEMM_SPLITK=$v timeout 200 python bench.py --config c10_b512 --steps 30 --warmup 5 --no-cpu-baseline --no-g-step --dev-local-batch 64 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])"; done
