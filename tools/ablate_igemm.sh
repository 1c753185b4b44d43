set -e
cd $GRAFT_REPO_ROOT
export CONV_LAYERS=1,2,3 CONV_ITERS=5
for v in "" "-DIGEMM_ABLATE_LOADS" "-DIGEMM_ABLATE_LOADS -DIGEMM_ABLATE_STORES" "-DIGEMM_ABLATE_LOADS -DIGEMM_ABLATE_STORES -DIGEMM_ABLATE_BARRIER"; do
  CONTRAD_EXTRA_HIPCC_FLAGS="$v" python contrad_amd/build.py --force > /dev/null 2>&1
  echo "== variant: [$v]"
  python tools/bench_conv.py 2>&1 | tail -3
done
