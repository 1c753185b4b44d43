#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_igemm_gpu.py -x -q 2>&1 | tail -2
export CONV_ITERS=10 CONV_WARM=3 CONV_BATCH=48
export CONV_CUSTOM="512,32,32,3,1,1;256,64,64,3,1,1;513,32,64,3,2,0;128,128,128,3,1,1"
timeout 120 python tools/bench_conv.py 2>&1 | grep "^H"
timeout 120 python tools/bench_conv.py 2>&1 | grep "^H"
