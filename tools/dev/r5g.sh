# round 5, call g: demodulation kernel v2 (wave-split Cin, 16 loads in flight), single-channel u2d1 planes kernel: parity,
# durations from a kernel trace, same-box A/B against the library of commit 23ed398 (before u2d1 / demod v2 / planes)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5g; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_stylegan2_gpu.py tests/test_stylegan2_512_gpu.py tests/test_stylegan2_gstep_gpu.py tests/test_graph_gpu.py -q -m gpu -x > $O/pytest.log 2>&1; grep -E "passed|failed" $O/pytest.log
OLD=$R/contrad_amd/csrc/variants/libcontrad_u2d1old.so
for rep in 1 2 3; do for L in "$OLD" "-"; do
  for c in sg2_512 sg2_32; do
    if [ "$L" = "-" ]; then unset CONTRAD_HIP_LIB; tag=new; else export CONTRAD_HIP_LIB=$L; tag=old; fi
    r=$(timeout 300 python bench.py --config $c --no-cpu-baseline --no-g-step 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])")
    echo "lib $c rep$rep [$tag] $r" | tee -a $O/ab.txt
  done
done; done
unset CONTRAD_HIP_LIB
cd /tmp && export TMPDIR=/tmp
for c in sg2_512 sg2_32; do
  timeout 300 rocprofv3 --kernel-trace -d $O -o ${c}_kt -- python $R/bench.py --config $c --steps 8 --warmup 2 --no-cpu-baseline --no-g-step > $O/${c}_kt.log 2>&1
  python $R/tools/rocpd_summary.py $O/${c}_kt_results.db > $O/${c}_kernel_trace.txt 2>&1; rm -f $O/${c}_kt_results.db
  grep -E "modconv|upfirdn|TOTAL" $O/${c}_kernel_trace.txt | cut -c1-140
done
