# round 6: F(2x2,2x2) weight gradient: parity + per-layer A/B (dev library)
R=$GRAFT_REPO_ROOT; cd $R
timeout 600 python -m pytest tests/test_wino_gpu.py -x -q 2>&1 | tail -5
export CONTRAD_HIP_LIB=$R/contrad_amd/csrc/libcontrad_hip_dev.so CONV_MODES=wgrad,dgrad
for w in 1 0 1 0; do
  echo "== CONTRAD_WINO22_WGRAD=$w"
  CONTRAD_WINO22_WGRAD=$w CONV_LAYERS=0,2,4 python tools/bench_conv.py 2>&1 | grep -v amdgpu.ids
done
