# Dev (round 4): class-group size of the plain strided order vs blocks per XCD round (32 CUs): CGROUP x tiles_n = 32 puts
# one block of every class on every CU of a single-round launch.  Output: gpurun_out/dgrad_cgroup2.txt
cd "$GRAFT_REPO_ROOT"
export CONTRAD_HIP_LIB=$PWD/contrad_amd/csrc/libcontrad_hip_dev.so CONV_MODES=dgrad CONV_ITERS=20 CONTRAD_DGRAD_BALANCE=0
OUT=gpurun_out/dgrad_cgroup2.txt; : > $OUT
SH="33,128,256,3,2,0;17,256,512,3,2,0;9,512,512,3,2,0;33,512,512,3,2,0;17,512,512,3,2,0;65,256,512,3,2,0;129,128,256,3,2,0"
for B in 16 48 64 192; do for G in 4 8 16 32 64; do for T in 0 128128 64128 64064; do
  echo "== batch $B cgroup $G tile $T" >> $OUT
  if [ $T = 0 ]; then unset CONTRAD_IGEMM_TILE; else export CONTRAD_IGEMM_TILE=$T; fi
  CONTRAD_DGRAD_CGROUP=$G CONV_BATCH=$B CONV_CUSTOM="$SH" timeout 300 python tools/bench_conv.py 2>&1 | grep "^H" | sed 's/| fwd.*| dgrad/| dgrad/; s/| wgrad.*//' >> $OUT
done; done; done
