R=$GRAFT_REPO_ROOT
cd $R
timeout 1500 python -m pytest tests -q -m gpu --timeout 600 -p no:cacheprovider 2>&1 | grep -a "passed\|failed\|FAILED" | tail -6
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
