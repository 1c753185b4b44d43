# round 6: dynamic item queue of wino_kernel: parity, per-layer times, overlap rehearsal
cd "$GRAFT_REPO_ROOT"
python -m pytest tests/test_wino_gpu.py -x -q 2>&1 | tail -2
CONV_MODES=fwd,dgrad CONV_LAYERS=1,3,5 python tools/bench_conv.py 2>&1 | grep "^H"
CONV_MODES=fwd,dgrad CONV_BATCH=48 CONV_CUSTOM="256,64,64,3,1,1;128,128,128,3,1,1;32,512,512,3,1,1" python tools/bench_conv.py 2>&1 | grep "^H"
python tools/overlap_rehearsal.py both 2>&1 | grep -v amdgpu.ids
