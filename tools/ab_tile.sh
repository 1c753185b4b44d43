# Dev: which block tile wins at small per-rank batches?  (CONTRAD_IGEMM_TILE forces the FWD / DGRAD tile.)
cd $GRAFT_REPO_ROOT
export CONV_ITERS=30 CONV_WARM=5
for B in 192 384; do
  export CONV_BATCH=$B
  for t in 0 128128 128064 64128 64064; do
    echo "== batch $B tile $t"
    CONTRAD_IGEMM_TILE=$t timeout 100 python tools/bench_conv.py 2>&1 | grep -v total | cut -c1-125
  done
done
