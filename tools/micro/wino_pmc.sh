# Dev: SQ / LDS counters of the Winograd prototype (tools/micro/wino_proto.bin).  usage: wino_pmc.sh [binary]
R=$GRAFT_REPO_ROOT
B=${1:-$R/tools/micro/wino_proto.bin}
O=$R/gpurun_out/wino_pmc
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 100 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU -d $O -o sq -- $B 3 one > $O/sq.log 2>&1
timeout 100 rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM -d $O -o lds -- $B 3 one > $O/lds.log 2>&1
timeout 100 rocprofv3 --kernel-trace --pmc SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_LEVEL_VMEM GRBM_GUI_ACTIVE -d $O -o misc -- $B 3 one > $O/misc.log 2>&1
for k in sq lds misc; do echo "== $k"; python $R/tools/rocpd_pmc.py $O/${k}_results.db wino; done
