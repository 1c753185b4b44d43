# round 6: whole-step A/B of the F(2x2,2x2) kernel (dev library) + the full GPU suite
cd "$GRAFT_REPO_ROOT"
for c in c10_b512 sg2_32; do bash tools/dev/ab_env.sh $c "CONTRAD_WINO22=1" "CONTRAD_WINO22=0"; done
python -m pytest tests -m gpu -x -q 2>&1 | tail -6
