R=$GRAFT_REPO_ROOT
cd $R
show() { python -c "
import sys,json; d=json.loads(sys.stdin.read())
print('$1', d['config']['name'], d['ms_per_step'], [(k, v['ms_per_step']) for k,v in d.get('other_configs',{}).items()])"; }
timeout 600 python bench.py 2>/dev/null | show "full+cpu (first)"
timeout 600 python bench.py --no-cpu-baseline 2>/dev/null | show "full nocpu"
timeout 600 python bench.py 2>/dev/null | show "full+cpu (again)"
timeout 600 python bench.py --config sg2_512 2>/dev/null | show "sg2_512+cpu"
