# Dev (round 4): strided 3x3 stride-2 data gradients, balanced block order (igemm.hip dgrad_balance) vs the class-group
# order, on the StyleGAN2 shapes.  Needs the dev build (switches compiled in).  Output: gpurun_out/dgrad_balance.txt
cd "$GRAFT_REPO_ROOT"
export CONTRAD_HIP_LIB=$PWD/contrad_amd/csrc/libcontrad_hip_dev.so CONV_MODES=dgrad CONV_ITERS=20
OUT=gpurun_out/dgrad_balance.txt; : > $OUT
SH512="513,32,64,3,2,0;257,64,128,3,2,0;129,128,256,3,2,0;65,256,512,3,2,0;33,512,512,3,2,0;17,512,512,3,2,0;9,512,512,3,2,0"
SH32="33,128,256,3,2,0;17,256,512,3,2,0;9,512,512,3,2,0"
for B in 48 16; do for BAL in 0 1; do
  echo "== batch $B balance $BAL" >> $OUT
  CONTRAD_DGRAD_BALANCE=$BAL CONV_BATCH=$B CONV_CUSTOM="$SH512" timeout 300 python tools/bench_conv.py 2>&1 | grep "^H" | sed 's/| fwd.*| dgrad/| dgrad/; s/| wgrad.*//' >> $OUT
done; done
for B in 192 64; do for BAL in 0 1; do
  echo "== batch $B balance $BAL" >> $OUT
  CONTRAD_DGRAD_BALANCE=$BAL CONV_BATCH=$B CONV_CUSTOM="$SH32" timeout 300 python tools/bench_conv.py 2>&1 | grep "^H" | sed 's/| fwd.*| dgrad/| dgrad/; s/| wgrad.*//' >> $OUT
done; done
for T in 128128 64128 128064 64064; do for B in 48 64 192; do
  echo "== batch $B balance 1 tile $T" >> $OUT
  S="$SH32"; [ $B = 48 ] && S="129,128,256,3,2,0;65,256,512,3,2,0;33,512,512,3,2,0;17,512,512,3,2,0"
  CONTRAD_IGEMM_TILE=$T CONTRAD_DGRAD_BALANCE=1 CONV_BATCH=$B CONV_CUSTOM="$S" timeout 300 python tools/bench_conv.py 2>&1 | grep "^H" | sed 's/| fwd.*| dgrad/| dgrad/; s/| wgrad.*//' >> $OUT
done; done
cat $OUT
