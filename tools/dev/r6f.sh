# round 6: whole-step numbers with the Winograd weight gradient + the full GPU suite
cd "$GRAFT_REPO_ROOT"
for c in c10_b512 sg2_512 sg2_32; do bash tools/dev/ab_env.sh $c "CONTRAD_WINO_WGRAD=1" "CONTRAD_WINO_WGRAD=0"; done
python -m pytest tests -m gpu -x -q 2>&1 | tail -6
