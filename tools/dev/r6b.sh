# round 6: kernel trace of the Winograd layers alone (filter transform vs main kernel)
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
O=$R/gpurun_out/r6b; rm -rf $O; mkdir -p $O
CONV_MODES=fwd,dgrad CONV_LAYERS=1,3,5 timeout 200 rocprofv3 --kernel-trace --stats -d $O -o sn -- python $R/tools/bench_conv.py > $O/sn.log 2>&1
cat $O/sn.log | grep -v amdgpu.ids
python $R/tools/rocpd_summary.py $O/sn_results.db 2>/dev/null | head -20 || ls $O
