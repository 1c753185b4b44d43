cd $GRAFT_REPO_ROOT
for rep in 1 2; do for L in old new; do
  if [ $L = old ]; then export CONTRAD_HIP_LIB=$PWD/contrad_amd/csrc/libcontrad_hip_old.so; else unset CONTRAD_HIP_LIB; fi
  echo "== $L"
  CONV_CUSTOM="32,64,128,4,2,1;16,128,256,4,2,1;8,256,512,4,2,1" python tools/bench_conv.py 2>&1 | grep "^H"
  CONV_BATCH=512 CONV_CUSTOM="32,64,128,4,2,1;16,128,256,4,2,1;8,256,512,4,2,1" python tools/bench_conv.py 2>&1 | grep "^H"
done; done
for rep in 1 2; do for L in old new; do
  if [ $L = old ]; then export CONTRAD_HIP_LIB=$PWD/contrad_amd/csrc/libcontrad_hip_old.so; else unset CONTRAD_HIP_LIB; fi
  R=$(timeout 600 python bench.py --config c10_b512 --no-cpu-baseline --no-g-step 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['value'])")
  echo "c10_b512 rep$rep [$L] ms/step, img/s: $R"
done; done
