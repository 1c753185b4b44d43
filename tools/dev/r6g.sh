# round 6: F(2x2,2x2) kernel for the 4x4 stride-2 layers: parity + per-layer A/B (dev library)
R=$GRAFT_REPO_ROOT; cd $R
timeout 600 python -m pytest tests/test_wino_gpu.py -x -q -k wino22 2>&1 | tail -12
export CONTRAD_HIP_LIB=$R/contrad_amd/csrc/libcontrad_hip_dev.so CONV_MODES=fwd,dgrad
for w in 1 0 1 0; do
  echo "== CONTRAD_WINO22=$w"
  CONTRAD_WINO22=$w CONV_LAYERS=0,2,4 python tools/bench_conv.py 2>&1 | grep -v amdgpu.ids
  CONTRAD_WINO22=$w CONV_BATCH=512 CONV_LAYERS=0,2,4 CONV_MODES=dgrad python tools/bench_conv.py 2>&1 | grep -v amdgpu.ids
done
