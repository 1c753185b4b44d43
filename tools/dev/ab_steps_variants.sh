# Dev: whole steps of bench.py under library variants (tools/build_variant.sh) on ONE box.  usage: ab_steps_variants.sh "configs" variant ... ("-" = shipped)
cd $GRAFT_REPO_ROOT
C=$1; shift
for c in $C; do for rep in 1 2; do for v in "$@"; do
  if [ "$v" = "-" ]; then unset CONTRAD_HIP_LIB; else export CONTRAD_HIP_LIB=$PWD/contrad_amd/csrc/variants/libcontrad_$v.so; fi
  R=$(timeout 600 python bench.py --config $c --no-cpu-baseline --no-g-step 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['value'])")
  echo "$c rep$rep [$v] ms/step, img/s: $R"
done; done; done
