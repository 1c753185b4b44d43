# round 6: plan thresholds of the Winograd kernels at per-rank batches (one rank of the 8- / 4-GPU headline config on one GPU) and on sg2_32
cd "$GRAFT_REPO_ROOT"
export CONTRAD_HIP_LIB=$PWD/contrad_amd/csrc/libcontrad_hip_dev.so
run() { env $1 timeout 300 python bench.py --config $2 $3 --no-cpu-baseline --no-g-step 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'])"; }
for rep in 1 2; do
for E in "CONTRAD_WINO_MIN_ITEMS=200 CONTRAD_WINO22_MIN_ITEMS=200 CONTRAD_WINO_MIN_QPS=48" "CONTRAD_WINO_MIN_ITEMS=150 CONTRAD_WINO22_MIN_ITEMS=150 CONTRAD_WINO_MIN_QPS=16" "CONTRAD_WINO_MIN_ITEMS=90 CONTRAD_WINO22_MIN_ITEMS=150 CONTRAD_WINO_MIN_QPS=16" "CONTRAD_WINO_MIN_ITEMS=150 CONTRAD_WINO22_MIN_ITEMS=90 CONTRAD_WINO_MIN_QPS=8" "CONTRAD_WINO=0 CONTRAD_WINO22=0"; do
  echo "rep$rep [$E] b64: $(run "$E" c10_b512 "--dev-local-batch 64 --force-dist --steps 100")  b128: $(run "$E" c10_b512 "--dev-local-batch 128 --force-dist --steps 60")  sg2_32: $(run "$E" sg2_32 "")"
done; done
