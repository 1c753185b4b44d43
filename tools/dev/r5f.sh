# round 5, call f: upfirdn4_u2d1 with compile-time tap indices (no promoted-to-LDS weight array): parity, same-box A/B against
# the library of the commit before, LDS counters + HBM bytes of the kernel
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5f; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_stylegan2_gpu.py tests/test_stylegan2_512_gpu.py tests/test_stylegan2_gstep_gpu.py tests/test_r1_gradient_gpu.py tests/test_fullsize_oracle_gpu.py -q -m gpu -x > $O/pytest.log 2>&1; grep -E "passed|failed" $O/pytest.log
OLD=$R/contrad_amd/csrc/variants/libcontrad_u2d1old.so
for rep in 1 2 3; do for L in "$OLD" "-"; do
  for c in sg2_512 sg2_32; do
    if [ "$L" = "-" ]; then unset CONTRAD_HIP_LIB; tag=new; else export CONTRAD_HIP_LIB=$L; tag=old; fi
    r=$(timeout 300 python bench.py --config $c --no-cpu-baseline --no-g-step 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])")
    echo "u2d1 $c rep$rep [$tag] $r" | tee -a $O/ab.txt
  done
done; done
unset CONTRAD_HIP_LIB
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --config sg2_512 --steps 2 --warmup 2 --no-cpu-baseline --no-g-step --graph off"
timeout 240 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $O -o misc -- $B > $O/misc.log 2>&1
python $R/tools/rocpd_pmc.py $O/misc_results.db upfirdn4 > $O/pmc_upfirdn.txt 2>&1; rm -f $O/*.db; cat $O/pmc_upfirdn.txt
