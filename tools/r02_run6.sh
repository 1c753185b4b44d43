R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02f
rm -rf $O; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -q -m gpu --timeout 600 -p no:cacheprovider > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
grep -n "^FAILED\|^ERROR\|passed\|failed" $O/pytest.log | head -40
