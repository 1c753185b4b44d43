# Dev: interleaved A/B of library variants on the default bench (ms per step).  args: variant names
cd $GRAFT_REPO_ROOT
for round in 1 2; do
  for v in base "$@"; do
    if [ "$v" = base ]; then unset CONTRAD_HIP_LIB; else export CONTRAD_HIP_LIB=$GRAFT_REPO_ROOT/contrad_amd/csrc/variants/libcontrad_$v.so; fi
    echo "== round $round variant $v: $(timeout 100 python bench.py --no-cpu-baseline 2>&1 | grep -o '"ms_per_step": [0-9.]*' | head -1)"
  done
done
