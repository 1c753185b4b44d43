# round 6: Winograd parity + per-layer times (shipped library)
R=$GRAFT_REPO_ROOT; cd $R
python -m pytest tests/test_wino_gpu.py -x -q 2>&1 | tail -3
export CONV_MODES=fwd,dgrad
for rep in 1 2; do
  CONV_LAYERS=1,3,5 python tools/bench_conv.py 2>&1 | grep -v amdgpu.ids
  CONV_BATCH=48 CONV_CUSTOM="256,64,64,3,1,1;128,128,128,3,1,1;64,256,256,3,1,1;32,512,512,3,1,1;16,512,512,3,1,1" python tools/bench_conv.py 2>&1 | grep -v amdgpu.ids
done
