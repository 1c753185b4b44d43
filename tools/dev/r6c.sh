# round 6: whole-step A/B Winograd on / off (dev library) on one box, all three workloads
cd "$GRAFT_REPO_ROOT"
for c in c10_b512 sg2_512 sg2_32; do bash tools/dev/ab_env.sh $c "CONTRAD_WINO=1" "CONTRAD_WINO=0"; done
