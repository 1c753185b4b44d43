# Dev: SQ / LDS counters of the F(4x4, 3x3) prototype (tools/micro/wino44_proto.bin; timing section only: "3 r s").  usage: wino44_pmc.sh [binary]
R=$GRAFT_REPO_ROOT
B=${1:-$R/tools/micro/wino44_proto.bin}
O=$R/gpurun_out/wino44_pmc
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 100 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU -d $O -o sq -- $B 3 r s > $O/sq.log 2>&1
timeout 100 rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM -d $O -o lds -- $B 3 r s > $O/lds.log 2>&1
timeout 100 rocprofv3 --kernel-trace --pmc SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INSTS_MFMA SQ_INST_LEVEL_VMEM GRBM_GUI_ACTIVE -d $O -o misc -- $B 3 r s > $O/misc.log 2>&1
timeout 100 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum -d $O -o tcc -- $B 3 r s > $O/tcc.log 2>&1
timeout 100 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O -o fetch -- $B 3 r s > $O/fetch.log 2>&1
for k in sq lds misc tcc fetch; do echo "== $k"; python $R/tools/rocpd_pmc.py $O/${k}_results.db wino44_kernel; done
