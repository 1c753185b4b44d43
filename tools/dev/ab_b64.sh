# Dev: A/B of the per-rank batch-64 step (one rank of the 8-GPU headline config on one GPU) under different environments
cd "$GRAFT_REPO_ROOT"
export CONTRAD_HIP_LIB=$PWD/contrad_amd/csrc/libcontrad_hip_dev.so
for rep in 1 2; do for E in "$@"; do
  [ "$E" = "-" ] && E=""
  R=$(env $E timeout 600 python bench.py --config c10_b512 --dev-local-batch 64 --force-dist --steps 200 --warmup 5 --no-cpu-baseline --no-g-step 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'])")
  echo "b64 rep$rep [$E] ms/step: $R"
done; done
