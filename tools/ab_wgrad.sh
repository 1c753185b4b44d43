cd $GRAFT_REPO_ROOT
export CONV_ITERS=30 CONV_WARM=5
for B in 192 384; do
  export CONV_BATCH=$B
  for cfg in "0 1024" "64064 1024" "64128 1024" "0 768" "0 512" "64064 512"; do
    set -- $cfg
    echo "== batch $B tile $1 blocks $2: $(CONTRAD_WGRAD_TILE=$1 CONTRAD_WGRAD_BLOCKS=$2 timeout 100 python tools/bench_conv.py 2>&1 | grep -v total | awk '{printf "%s ", $(NF-3)}')"
  done
done
