# round 5, call h: non-temporal loads (1) / stores (2) / both (3) in the FIR kernels: isolated kernel rates + sg2_512 step A/B
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5h; mkdir -p $O; cd $R
timeout 300 python -m pytest tests/test_stylegan2_gpu.py -q -m gpu -x -k "fir4 or upfirdn" > $O/pytest.log 2>&1; grep -E "passed|failed" $O/pytest.log
for v in base nt1 nt2 nt3; do
  if [ "$v" = base ]; then unset CONTRAD_HIP_LIB; else export CONTRAD_HIP_LIB=$R/contrad_amd/csrc/variants/libcontrad_$v.so; fi
  echo "== $v" | tee -a $O/hbm.txt
  timeout 200 python tools/bench_hbm.py sg2 2>&1 | grep -E "upfirdn|lincomb" | tee -a $O/hbm.txt
done
for rep in 1 2; do for v in base nt1 nt2 nt3; do
  if [ "$v" = base ]; then unset CONTRAD_HIP_LIB; else export CONTRAD_HIP_LIB=$R/contrad_amd/csrc/variants/libcontrad_$v.so; fi
  r=$(timeout 300 python bench.py --config sg2_512 --no-cpu-baseline --no-g-step 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])")
  echo "nt sg2_512 rep$rep [$v] $r" | tee -a $O/ab.txt
done; done
