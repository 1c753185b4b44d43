# Dev: interleaved A/B of library variants (tools/build_variant.sh) on the conv micro-benchmark.  args: variant names
cd $GRAFT_REPO_ROOT
export CONV_ITERS=${CONV_ITERS:-20} CONV_WARM=5
for round in 1 2; do
  for v in base "$@"; do
    if [ "$v" = base ]; then unset CONTRAD_HIP_LIB; else export CONTRAD_HIP_LIB=$GRAFT_REPO_ROOT/contrad_amd/csrc/variants/libcontrad_$v.so; fi
    echo "== round $round variant $v: $(timeout 100 python tools/bench_conv.py 2>&1 | grep total | awk '{printf "%s %s TF  ", $1, $5}')"
  done
done
