# Dev: tools/bench_conv.py under two libraries on ONE box.  usage: ab_conv.sh LIB_A LIB_B   ("-" = the in-tree library);
# CONV_CUSTOM / CONV_MODES / CONV_BATCH as for bench_conv.py
cd "$GRAFT_REPO_ROOT"
for rep in 1 2; do for L in "$@"; do
  if [ "$L" = "-" ]; then unset CONTRAD_HIP_LIB; else export CONTRAD_HIP_LIB=$PWD/$L; fi
  echo "== rep$rep [$L]"; python tools/bench_conv.py 2>&1 | grep "^H" | cut -c1-150
done; done
