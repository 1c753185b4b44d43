# round 5, call k: non-temporal activation stores chosen by output size (>= 256 MB): full GPU suite, same-box A/B against
# "never" and "always"
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5k; mkdir -p $O; cd $R
V=$R/contrad_amd/csrc/variants
timeout 900 python -m pytest tests -q -m gpu -x > $O/pytest.log 2>&1; grep -E "passed|failed" $O/pytest.log
CONTRAD_HIP_LIB=$V/libcontrad_stall.so timeout 400 python -m pytest tests/test_igemm_gpu.py tests/test_kernels_gpu.py -q -m gpu -x > $O/pytest_stall.log 2>&1; grep -E "passed|failed" $O/pytest_stall.log
run() { # cfg tag lib
  if [ "$3" = "-" ]; then unset CONTRAD_HIP_LIB; else export CONTRAD_HIP_LIB=$V/libcontrad_$3.so; fi
  r=$(timeout 300 python bench.py --config $1 --no-cpu-baseline --no-g-step 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])")
  echo "st $1 [$2] $r" | tee -a $O/ab.txt
}
for rep in 1 2 3; do for c in sg2_512 c10_b512 sg2_32; do run $c never stoff; run $c by-size -; run $c always stall; done; done
