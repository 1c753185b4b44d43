#!/bin/bash
# dev: strided 3x3 data gradients, one block per (tile, class) vs all classes chained in one block
cd $GRAFT_REPO_ROOT
export CONV_ITERS=20 CONV_WARM=5
for spec in "64:33,128,256,3,2,0;17,256,512,3,2,0;9,512,512,3,2,0" "192:33,128,256,3,2,0;17,256,512,3,2,0;9,512,512,3,2,0" "16:513,32,64,3,2,0;257,64,128,3,2,0;129,128,256,3,2,0;65,256,512,3,2,0;33,512,512,3,2,0;17,512,512,3,2,0" "48:513,32,64,3,2,0;257,64,128,3,2,0;129,128,256,3,2,0;65,256,512,3,2,0;33,512,512,3,2,0;17,512,512,3,2,0"; do
  export CONV_BATCH=${spec%%:*} CONV_CUSTOM="${spec#*:}"
  for c in 0 1; do
    export CONTRAD_DGRAD_CHAIN=$c
    echo "== batch $CONV_BATCH chain $c"
    timeout 100 python tools/bench_conv.py 2>&1 | grep "^H" | sed 's/| fwd.*| dgrad/| dgrad/; s/| wgrad.*//'
  done
done
