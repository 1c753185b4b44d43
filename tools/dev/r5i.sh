# round 5, call i: non-temporal stores -- FIR kernels by output size (never / >= 128 MB (default) / >= 32 MB / always) and
# the conv engine's activation stores (LEAN_ST_AUX = 2): parity, same-box A/B on all workloads, L2 hit rate + traffic
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5i; mkdir -p $O; cd $R
V=$R/contrad_amd/csrc/variants
CONTRAD_HIP_LIB=$V/libcontrad_stnt.so timeout 400 python -m pytest tests/test_igemm_gpu.py tests/test_kernels_gpu.py -q -m gpu -x > $O/pytest_stnt.log 2>&1; grep -E "passed|failed" $O/pytest_stnt.log
timeout 300 python -m pytest tests/test_stylegan2_gpu.py tests/test_stylegan2_512_gpu.py -q -m gpu -x > $O/pytest.log 2>&1; grep -E "passed|failed" $O/pytest.log
run() { # cfg tag lib
  if [ "$3" = "-" ]; then unset CONTRAD_HIP_LIB; else export CONTRAD_HIP_LIB=$V/libcontrad_$3.so; fi
  r=$(timeout 300 python bench.py --config $1 --no-cpu-baseline --no-g-step 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])")
  echo "nt $1 [$2] $r" | tee -a $O/ab.txt
}
for rep in 1 2; do
  for c in sg2_512 sg2_32; do run $c never ntnever; run $c nt128 -; run $c nt32 nt32; run $c always ntalways; run $c igemm-stnt stnt; done
  run c10_b512 base -; run c10_b512 igemm-stnt stnt
done
unset CONTRAD_HIP_LIB
cd /tmp && export TMPDIR=/tmp
for v in base stnt; do
  if [ "$v" = base ]; then unset CONTRAD_HIP_LIB; else export CONTRAD_HIP_LIB=$V/libcontrad_$v.so; fi
  B="python $R/bench.py --config c10_b512 --steps 3 --warmup 2 --no-cpu-baseline --no-g-step --graph off"
  timeout 200 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum -d $O -o tcc_$v -- $B > $O/tcc_$v.log 2>&1
  timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O -o fetch_$v -- $B > $O/fetch_$v.log 2>&1
  echo "== $v" >> $O/pmc.txt
  python $R/tools/rocpd_pmc.py $O/tcc_${v}_results.db igemm_lean >> $O/pmc.txt 2>&1
  python $R/tools/rocpd_pmc.py $O/fetch_${v}_results.db igemm_lean >> $O/pmc.txt 2>&1
done
rm -f $O/*.db; cat $O/pmc.txt | grep -v "^$" | head -80
