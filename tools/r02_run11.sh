R=$GRAFT_REPO_ROOT
cd $R
timeout 1500 python -m pytest tests -q -m gpu --timeout 600 -p no:cacheprovider 2>&1 | grep -a "passed\|failed\|FAILED" | tail -8
echo "--- bench --gpus 2 on a 1-GPU box (expected: launcher starts 2 ranks, rank 1 fails on its device)"
timeout 120 python bench.py --gpus 2 --config c10_b512 --steps 2 --warmup 1 --no-cpu-baseline > /tmp/g2.out 2> /tmp/g2.err; echo "rc=$?"; tail -c 600 /tmp/g2.err | tr '\n' ' ' | cut -c1-600; echo; head -c 300 /tmp/g2.out
