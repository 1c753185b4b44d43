# Dev: A/B of bench.py under different environments on ONE box (box-to-box spread is +-3 %).
# usage: ab_env.sh CONFIG "ENV1" "ENV2" ...   (ENV = "VAR=val VAR2=val"; "-" = no variables); uses the dev library
cd "$GRAFT_REPO_ROOT"
CFG=$1; shift
export CONTRAD_HIP_LIB=$PWD/contrad_amd/csrc/libcontrad_hip_dev.so
for rep in 1 2; do for E in "$@"; do
  [ "$E" = "-" ] && E=""
  R=$(env $E timeout 600 python bench.py --config $CFG --no-cpu-baseline --no-g-step 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['value'])")
  echo "$CFG rep$rep [$E] ms/step, img/s: $R"
done; done
