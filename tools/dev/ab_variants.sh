# Dev: per-layer times of tools/bench_conv.py under library variants (tools/build_variant.sh) on ONE box.
# usage: ab_variants.sh "CONV_CUSTOM list" variant1 variant2 ...   ("-" = the shipped library)
cd $GRAFT_REPO_ROOT
L=$1; shift
for rep in 1 2; do for v in "$@"; do
  if [ "$v" = "-" ]; then unset CONTRAD_HIP_LIB; else export CONTRAD_HIP_LIB=$PWD/contrad_amd/csrc/variants/libcontrad_$v.so; fi
  echo "== $v"; CONV_CUSTOM="$L" python tools/bench_conv.py 2>&1 | grep "^H"
done; done
