set -e
cd $GRAFT_REPO_ROOT
export CONV_ITERS=5
for v in "-DIGEMM_BK=32 -DIGEMM_MIN_WAVES=2" "-DIGEMM_BK=16 -DIGEMM_MIN_WAVES=2" "-DIGEMM_BK=16 -DIGEMM_MIN_WAVES=4" "-DIGEMM_BK=16 -DIGEMM_MIN_WAVES=3"; do
  CONTRAD_EXTRA_HIPCC_FLAGS="$v" python contrad_amd/build.py --force > /tmp/b.log 2>&1 || { tail -5 /tmp/b.log; continue; }
  echo "== variant: [$v]"
  python -m pytest tests/test_igemm_gpu.py -m gpu -q --no-header -p no:cacheprovider -x 2>&1 | tail -1
  python tools/bench_conv.py 2>&1 | tail -10
done
python contrad_amd/build.py --force > /dev/null 2>&1
