# round 5, call j: non-temporal activation stores in the conv engine's epilogues (LEAN_ST_AUX = 2, conv_c32 too): parity,
# same-box A/B on all workloads, L2 hit rate + FETCH_SIZE of the headline convs
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5j; mkdir -p $O; cd $R
V=$R/contrad_amd/csrc/variants
CONTRAD_HIP_LIB=$V/libcontrad_stnt.so timeout 400 python -m pytest tests/test_igemm_gpu.py tests/test_kernels_gpu.py tests/test_sndcgan_gpu.py -q -m gpu -x > $O/pytest_stnt.log 2>&1; grep -E "passed|failed" $O/pytest_stnt.log
run() { # cfg tag lib
  if [ "$3" = "-" ]; then unset CONTRAD_HIP_LIB; else export CONTRAD_HIP_LIB=$V/libcontrad_$3.so; fi
  r=$(timeout 300 python bench.py --config $1 --no-cpu-baseline --no-g-step 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])")
  echo "stnt $1 [$2] $r" | tee -a $O/ab.txt
}
for rep in 1 2 3; do for c in c10_b512 sg2_512 sg2_32; do run $c base -; run $c igemm-stnt stnt; done; done
unset CONTRAD_HIP_LIB
cd /tmp && export TMPDIR=/tmp
for v in base stnt; do
  if [ "$v" = base ]; then unset CONTRAD_HIP_LIB; else export CONTRAD_HIP_LIB=$V/libcontrad_$v.so; fi
  B="python $R/bench.py --config c10_b512 --steps 3 --warmup 2 --no-cpu-baseline --no-g-step --graph off"
  timeout 200 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum -d $O -o tcc_$v -- $B > $O/tcc_$v.log 2>&1
  timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O -o fetch_$v -- $B > $O/fetch_$v.log 2>&1
  echo "== $v" >> $O/pmc.txt
  python $R/tools/rocpd_pmc.py $O/tcc_${v}_results.db "igemm_lean_kernel<" >> $O/pmc.txt 2>&1
  python $R/tools/rocpd_pmc.py $O/fetch_${v}_results.db "igemm_lean_kernel<" >> $O/pmc.txt 2>&1
done
rm -f $O/*.db; grep -E "==|128, 128|TCC|FETCH" $O/pmc.txt | head -60
