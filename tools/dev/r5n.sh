# round 5, call n: the 32-channel kernels walking DOWN columns of tiles (vertical halo rows re-read at once): parity, five
# alternations on sg2_512, traffic + L2 hit rate of the three kernels
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5n; mkdir -p $O; cd $R
V=$R/contrad_amd/csrc/variants
CONTRAD_HIP_LIB=$V/libcontrad_c32y.so timeout 400 python -m pytest tests/test_igemm_gpu.py tests/test_stylegan2_512_gpu.py -q -m gpu -x > $O/pytest.log 2>&1; grep -E "passed|failed" $O/pytest.log
run() { if [ "$3" = "-" ]; then unset CONTRAD_HIP_LIB; else export CONTRAD_HIP_LIB=$V/libcontrad_$3.so; fi
  r=$(timeout 300 python bench.py --config $1 --no-cpu-baseline --no-g-step 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])")
  echo "n $1 [$2] $r" | tee -a $O/ab.txt; }
for rep in 1 2 3 4 5; do run sg2_512 rows -; run sg2_512 columns c32y; done
unset CONTRAD_HIP_LIB
cd /tmp && export TMPDIR=/tmp
for v in base c32y; do
  if [ "$v" = base ]; then unset CONTRAD_HIP_LIB; else export CONTRAD_HIP_LIB=$V/libcontrad_$v.so; fi
  B="python $R/bench.py --config sg2_512 --steps 2 --warmup 2 --no-cpu-baseline --no-g-step --graph off"
  timeout 240 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum -d $O -o tcc_$v -- $B > $O/tcc_$v.log 2>&1
  timeout 240 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O -o fetch_$v -- $B > $O/fetch_$v.log 2>&1
  echo "== $v" >> $O/pmc.txt
  python $R/tools/rocpd_pmc.py $O/tcc_${v}_results.db "c32" >> $O/pmc.txt 2>&1
  python $R/tools/rocpd_pmc.py $O/fetch_${v}_results.db "c32" >> $O/pmc.txt 2>&1
done
rm -f $O/*.db; cat $O/pmc.txt
