# round 6: Winograd weight gradient: parity + per-layer A/B (dev library)
R=$GRAFT_REPO_ROOT; cd $R
timeout 600 python -m pytest tests/test_wino_gpu.py -x -q 2>&1 | tail -12
export CONTRAD_HIP_LIB=$R/contrad_amd/csrc/libcontrad_hip_dev.so CONV_MODES=wgrad,dgrad
for w in 1 0; do
  echo "== CONTRAD_WINO_WGRAD=$w"
  CONTRAD_WINO_WGRAD=$w CONV_LAYERS=1,3,5 python tools/bench_conv.py 2>&1 | grep -v amdgpu.ids
  CONTRAD_WINO_WGRAD=$w CONV_BATCH=48 CONV_CUSTOM="256,64,64,3,1,1;128,128,128,3,1,1;64,256,256,3,1,1;32,512,512,3,1,1;16,512,512,3,1,1" python tools/bench_conv.py 2>&1 | grep -v amdgpu.ids
done
