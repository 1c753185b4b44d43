#!/bin/bash
# Does the second (partial) dispatch round of a pixel-major launch cost matrix-pipe time?  The same layers at image counts
# that give 0.75 / 1 / 1.5 / 2 rounds of 1024 resident blocks (tools/bench_conv.py prints ms and nominal TF/s per direction).
export CONV_CUSTOM="8,256,256,3,1,1;16,128,256,4,2,1;16,128,128,3,1,1"
export CONV_MODES=${CONV_MODES:-fwd,dgrad} CONV_ITERS=20
for B in 768 1024 1536 2048; do
  echo "== batch $B"; CONV_BATCH=$B python tools/bench_conv.py 2>&1 | grep -v "^$" | cut -c1-200
done
