# Dev: A/B of bench.py between two builds of the library on ONE box: ab_lib.sh CONFIG LIB_A LIB_B   ("-" = the shipped library)
cd "$GRAFT_REPO_ROOT"
CFG=$1; shift
for rep in 1 2; do for L in "$@"; do
  if [ "$L" = "-" ]; then unset CONTRAD_HIP_LIB; else export CONTRAD_HIP_LIB=$PWD/$L; fi
  R=$(timeout 600 python bench.py --config $CFG --no-cpu-baseline --no-g-step 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['value'])")
  echo "$CFG rep$rep [$L] ms/step, img/s: $R"
done; done
