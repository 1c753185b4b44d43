# Dev (round 6): the 4x4-FIR kernels with each XCD on a contiguous run of blocks (uf_block<true> in csrc/stylegan2_ops.hip) against launch order:
# ab_fir.sh VARIANT_LIB -- the shipped library against a variant built the other way (tools/build_variant.sh NAME "" stylegan2_ops after the edit): kernels alone, then whole steps.
cd "$GRAFT_REPO_ROOT"
V=${1:-contrad_amd/csrc/variants/libcontrad_ufxcd0.so}
for rep in 1 2; do
  echo "== shipped (UF_XCD=1)"; python tools/bench_hbm.py sg2 2>&1 | grep upfirdn
  echo "== UF_XCD=0"; CONTRAD_HIP_LIB=$PWD/$V python tools/bench_hbm.py sg2 2>&1 | grep upfirdn
done
bash tools/dev/ab_lib.sh sg2_512 - $V
bash tools/dev/ab_lib.sh sg2_512 - $V
bash tools/dev/ab_lib.sh sg2_32 - $V
