# Dev: LDS bank-conflict counters of one conv layer for library variants.  args: variant names ("base" = the real one)
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_lds
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export CONV_ITERS=2 CONV_WARM=1 CONV_LAYERS=${CONV_LAYERS:-3}
for v in "$@"; do
  if [ "$v" = base ]; then unset CONTRAD_HIP_LIB; else export CONTRAD_HIP_LIB=$R/contrad_amd/csrc/variants/libcontrad_$v.so; fi
  timeout 100 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS -d $O -o $v -- python $R/tools/bench_conv.py > $O/$v.log 2>&1
  echo "== $v"; python $R/tools/rocpd_pmc.py $O/${v}_results.db igemm_lean
done
