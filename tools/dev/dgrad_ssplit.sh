#!/bin/bash
# dev: K-splits per parity class for the stride-2 3x3 data gradients (CONTRAD_DGRAD_SSPLIT forces S; 0 = unsplit), dgrad column
# NEEDS tools/experiments/dgrad_strided_splitk.diff applied and the library rebuilt (the experiment was not adopted)
# of bench_conv (includes the slab reduce)
cd $GRAFT_REPO_ROOT
export CONV_ITERS=20 CONV_WARM=5
for spec in "64:33,128,256,3,2,0;17,256,512,3,2,0;9,512,512,3,2,0" "192:33,128,256,3,2,0;17,256,512,3,2,0;9,512,512,3,2,0" "16:513,32,64,3,2,0;257,64,128,3,2,0;129,128,256,3,2,0;65,256,512,3,2,0;33,512,512,3,2,0;17,512,512,3,2,0" "48:257,64,128,3,2,0;129,128,256,3,2,0;65,256,512,3,2,0;33,512,512,3,2,0;17,512,512,3,2,0"; do
  export CONV_BATCH=${spec%%:*} CONV_CUSTOM="${spec#*:}"
  for cfg in ${SSPLIT_CFGS-0:default 2:default 4:default 4:128128 4:64128 8:128128}; do
    export CONTRAD_DGRAD_SSPLIT=${cfg%%:*}
    t=${cfg#*:}
    if [ $t = default ]; then unset CONTRAD_IGEMM_TILE; else export CONTRAD_IGEMM_TILE=$t; fi
    echo "== batch $CONV_BATCH S $CONTRAD_DGRAD_SSPLIT tile $t"
    timeout 100 python tools/bench_conv.py 2>&1 | grep "^H" | sed 's/| fwd.*| dgrad/| dgrad/; s/| wgrad.*//'
  done
done
