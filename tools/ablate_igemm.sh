# Dev tool: where does the igemm main loop lose time?  Rebuilds the library with parts of the loop compiled out
# (results are WRONG in the ablated builds -- timing only) and runs the per-layer microbench.
#   r01 findings (3 mid layers, 116/103/116 GFLOP each):  full 112 TF/s | no global loads 129 | no loads, no LDS
#   stores 132 | + no barrier 133  => the barrier is free, LDS stores cost ~2 %, the global-load path ~13 %
#   (a two-deep register prefetch with 3x the latency slack did NOT recover it -> not a latency stall; the
#   ablated builds also clock higher because they draw less power).
set -e
cd $GRAFT_REPO_ROOT
export CONV_LAYERS=1,2,3 CONV_ITERS=5
for v in "" "-DIGEMM_ABLATE_LOADS" "-DIGEMM_ABLATE_LOADS -DIGEMM_ABLATE_STORES" "-DIGEMM_ABLATE_LOADS -DIGEMM_ABLATE_STORES -DIGEMM_ABLATE_BARRIER"; do
  CONTRAD_EXTRA_HIPCC_FLAGS="$v" python contrad_amd/build.py --force > /dev/null 2>&1
  echo "== variant: [$v]"
  python tools/bench_conv.py 2>&1 | tail -3
done
python contrad_amd/build.py --force > /dev/null 2>&1
