# Dev (round 6): the Winograd kernels against the direct kernels on ONE box (dev library switches).
#   wino_ab.sh layers   per-layer times (tools/bench_conv.py), 3x3 stride-1 and 4x4 stride-2 layers of SNDCGAN / StyleGAN2_512
#   wino_ab.sh steps    whole steps of the three bench workloads
#   wino_ab.sh plans    plan thresholds at per-rank batches (CONTRAD_WINO_MIN_ITEMS / _WINO22_MIN_ITEMS / _WINO_MIN_QPS)
#   wino_ab.sh wino44   F(4x4,3x3) against F(2x2,3x3) (CONTRAD_WINO44 = 1 | 0): the 3x3 layers alone, then whole steps
R=$GRAFT_REPO_ROOT; cd $R
export CONTRAD_HIP_LIB=$R/contrad_amd/csrc/libcontrad_hip_dev.so
case ${1:-layers} in
layers)
  for w in 1 0 1 0; do
    echo "== CONTRAD_WINO=$w CONTRAD_WINO22=$w"
    CONTRAD_WINO=$w CONTRAD_WINO22=$w python tools/bench_conv.py 2>&1 | grep "^H"
    CONTRAD_WINO=$w CONTRAD_WINO22=$w CONV_BATCH=48 CONV_CUSTOM="256,64,64,3,1,1;128,128,128,3,1,1;64,256,256,3,1,1;32,512,512,3,1,1;16,512,512,3,1,1" python tools/bench_conv.py 2>&1 | grep "^H"
  done;;
steps)
  for c in c10_b512 sg2_512 sg2_32; do bash tools/dev/ab_env.sh $c "CONTRAD_WINO=1" "CONTRAD_WINO=0 CONTRAD_WINO22=0"; done;;
plans)
  run() { env $1 timeout 300 python bench.py --config $2 $3 --no-cpu-baseline --no-g-step 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'])"; }
  for rep in 1 2; do for E in "CONTRAD_WINO_MIN_ITEMS=200 CONTRAD_WINO22_MIN_ITEMS=200 CONTRAD_WINO_MIN_QPS=48" "CONTRAD_WINO_MIN_ITEMS=150 CONTRAD_WINO22_MIN_ITEMS=150 CONTRAD_WINO_MIN_QPS=16" "CONTRAD_WINO=0 CONTRAD_WINO22=0"; do
    echo "rep$rep [$E] b64: $(run "$E" c10_b512 "--dev-local-batch 64 --force-dist --steps 100")  b128: $(run "$E" c10_b512 "--dev-local-batch 128 --force-dist --steps 60")  sg2_32: $(run "$E" sg2_32 "")"
  done; done;;
wino44)
  L3="16,128,128,3,1,1;8,256,256,3,1,1"
  L48="256,64,64,3,1,1;128,128,128,3,1,1;64,256,256,3,1,1;32,512,512,3,1,1"
  for w in 1 0 1 0; do
    echo "== CONTRAD_WINO44=$w (1536 images, then 48, then 16)"
    CONTRAD_WINO44=$w CONV_CUSTOM="$L3" python tools/bench_conv.py 2>&1 | grep "^H"
    CONTRAD_WINO44=$w CONV_BATCH=48 CONV_CUSTOM="$L48" python tools/bench_conv.py 2>&1 | grep "^H"
    CONTRAD_WINO44=$w CONV_BATCH=16 CONV_CUSTOM="$L48" python tools/bench_conv.py 2>&1 | grep "^H"
  done
  for c in c10_b512 sg2_512 sg2_32; do bash tools/dev/ab_env.sh $c "CONTRAD_WINO44=1" "CONTRAD_WINO44=0"; done;;
esac
