R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02v
rm -rf $O; mkdir -p $O
cd $R
for i in 1 2 3; do
timeout 900 python bench.py > $O/bench_$i.json 2> $O/bench_$i.err; echo "bench rc=$?"
python -c "
import json; d=json.load(open('$O/bench_$i.json'))
print('run $i', d['ms_per_step'], [(k, v['ms_per_step']) for k,v in d.get('other_configs',{}).items()])"
done
