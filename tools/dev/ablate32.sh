#!/bin/bash
# dev: which part of the lean loop holds the 32-channel layers back?  (ablated builds give WRONG results: timing only)
cd $GRAFT_REPO_ROOT
export CONV_ITERS=10 CONV_WARM=3 CONV_BATCH=48
export CONV_CUSTOM="512,32,32,3,1,1;256,64,64,3,1,1;513,32,64,3,2,0"
for v in base noload noload_nostore nofrag nothing; do
  echo "== $v"
  if [ $v = base ]; then unset CONTRAD_HIP_LIB; else export CONTRAD_HIP_LIB=$GRAFT_REPO_ROOT/contrad_amd/csrc/variants/libcontrad_$v.so; fi
  timeout 120 python tools/bench_conv.py 2>&1 | grep "^H"
done
