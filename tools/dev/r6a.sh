# round 6: Winograd parity + per-layer A/B against the direct kernels (dev library, CONTRAD_WINO=0)
R=$GRAFT_REPO_ROOT; cd $R
python -m pytest tests/test_wino_gpu.py -x -q 2>&1 | tail -15
export CONTRAD_HIP_LIB=$R/contrad_amd/csrc/libcontrad_hip_dev.so CONV_MODES=fwd,dgrad
for rep in 1 2; do
for w in 1 0; do
  echo "== CONTRAD_WINO=$w  SNDCGAN 3x3 layers, 1536 images"
  CONTRAD_WINO=$w CONV_LAYERS=1,3,5 python tools/bench_conv.py
  echo "== CONTRAD_WINO=$w  StyleGAN2_512 3x3 s1 layers, 48 images"
  CONTRAD_WINO=$w CONV_BATCH=48 CONV_CUSTOM="256,64,64,3,1,1;128,128,128,3,1,1;64,256,256,3,1,1;32,512,512,3,1,1;16,512,512,3,1,1;8,512,512,3,1,1" python tools/bench_conv.py
done
done
