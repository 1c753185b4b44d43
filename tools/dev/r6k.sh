# round 6: barrier before the last plane's MFMAs (default) against the barrier at the end of the chunk (variant lateb)
cd "$GRAFT_REPO_ROOT"
python -m pytest tests/test_wino_gpu.py -x -q 2>&1 | tail -2
export CONV_MODES=fwd,dgrad CONV_BATCH=48 CONV_CUSTOM="256,64,64,3,1,1;128,128,128,3,1,1;64,256,256,3,1,1;32,512,512,3,1,1"
bash tools/dev/ab_conv.sh - contrad_amd/csrc/variants/libcontrad_lateb.so
export CONV_BATCH=1536 CONV_CUSTOM="16,128,128,3,1,1;8,256,256,3,1,1;4,512,512,3,1,1"
bash tools/dev/ab_conv.sh - contrad_amd/csrc/variants/libcontrad_lateb.so
