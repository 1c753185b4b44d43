# round 6: whole-step numbers (shipped library) + the full GPU suite
cd "$GRAFT_REPO_ROOT"
for c in c10_b512 sg2_32 sg2_512; do
  R=$(timeout 600 python bench.py --config $c --no-cpu-baseline --no-g-step 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['value'])")
  echo "$c ms/step, img/s: $R"
done
python -m pytest tests -m gpu -x -q 2>&1 | tail -6
