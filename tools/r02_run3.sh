R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02c
rm -rf $O; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_stylegan2_gstep_gpu.py tests/test_stylegan2_512_gpu.py tests/test_augment_edge_gpu.py tests/test_dp_two_ranks_gpu.py tests/test_gstep_gpu.py -q -m gpu --timeout 600 -p no:cacheprovider > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
grep -n "^FAILED\|^ERROR\|passed\|failed" $O/pytest.log | head -40
