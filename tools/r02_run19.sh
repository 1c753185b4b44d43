R=$GRAFT_REPO_ROOT
cd $R
for g in on off; do timeout 600 python bench.py --no-cpu-baseline --graph $g 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read())
print('graph $g:', d['config']['name'], d['ms_per_step'], [(k, v['ms_per_step'], v['config']['launch']) for k,v in d['other_configs'].items()])"; done
