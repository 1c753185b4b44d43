# Dev tool: sample rocm-smi clocks / power while one conv layer runs in a sustained loop (args: layer index, iterations).
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
L=${1:-0}; IT=${2:-300}
( for i in $(seq 1 40); do timeout 5 rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Socket Power|mclk" | tr '\n' ' '; echo; sleep 0.15; done ) > $O/clock_$L.log &
SMI=$!
CONV_LAYERS=$L CONV_ITERS=$IT CONV_WARM=3 timeout 120 python $R/tools/bench_conv.py 2>&1 | tail -4
wait $SMI
sort $O/clock_$L.log | uniq -c | sort -rn | head -12
