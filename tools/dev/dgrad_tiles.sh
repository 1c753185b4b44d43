#!/bin/bash
# dev: which tile serves the stride-2 3x3 data gradients best?  (forced tile via CONTRAD_IGEMM_TILE, dgrad column of bench_conv)
cd $GRAFT_REPO_ROOT
export CONV_ITERS=20 CONV_WARM=5
for spec in "64:33,128,256,3,2,0;17,256,512,3,2,0;9,512,512,3,2,0" "192:33,128,256,3,2,0;17,256,512,3,2,0;9,512,512,3,2,0" "16:513,32,64,3,2,0;257,64,128,3,2,0;129,128,256,3,2,0;65,256,512,3,2,0;33,512,512,3,2,0;17,512,512,3,2,0" "48:65,256,512,3,2,0;33,512,512,3,2,0;17,512,512,3,2,0"; do
  export CONV_BATCH=${spec%%:*} CONV_CUSTOM="${spec#*:}"
  for t in default 128128 64128 128064 64064; do
    if [ $t = default ]; then unset CONTRAD_IGEMM_TILE; else export CONTRAD_IGEMM_TILE=$t; fi
    echo "== batch $CONV_BATCH tile $t"
    timeout 100 python tools/bench_conv.py 2>&1 | grep "^H" | sed 's/| fwd.*| dgrad/| dgrad/; s/| wgrad.*//'
  done
done
