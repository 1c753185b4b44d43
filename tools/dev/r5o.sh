# round 5, call o: two K-tiles of global loads in flight in the narrow instances (LEAN_DEEP = 1: WGRAD <= 128 x 64; 3: also
# FWD / DGRAD <= 128 x 64): parity, alternations on every workload
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5o; mkdir -p $O; cd $R
V=$R/contrad_amd/csrc/variants
for v in deep1 deep3; do CONTRAD_HIP_LIB=$V/libcontrad_$v.so timeout 400 python -m pytest tests/test_igemm_gpu.py tests/test_kernels_gpu.py -q -m gpu -x > $O/pytest_$v.log 2>&1; echo $v $(grep -E "passed|failed" $O/pytest_$v.log); done
run() { if [ "$3" = "-" ]; then unset CONTRAD_HIP_LIB; else export CONTRAD_HIP_LIB=$V/libcontrad_$3.so; fi
  r=$(timeout 300 python bench.py --config $1 --no-cpu-baseline --no-g-step 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])")
  echo "o $1 [$2] $r" | tee -a $O/ab.txt; }
for rep in 1 2 3 4 5; do run sg2_512 base -; run sg2_512 deep-wgrad deep1; run sg2_512 deep-all deep3; done
for rep in 1 2 3; do run sg2_32 base -; run sg2_32 deep-wgrad deep1; run sg2_32 deep-all deep3; done
for rep in 1 2 3; do run c10_b512 base -; run c10_b512 deep-all deep3; done
