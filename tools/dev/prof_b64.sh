# Dev: kernel trace + per-shape rows of one rank of the 8-GPU headline config (per-rank batch 64) on one GPU
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/b64p; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace -d $O -o kt -- python $R/bench.py --config c10_b512 --dev-local-batch 64 --force-dist --steps 20 --warmup 3 --no-cpu-baseline --no-g-step --shape-table $O/shapes.json > $O/kt.log 2>&1
python $R/tools/rocpd_summary.py $O/kt_results.db > $O/kernel_trace.txt 2>&1
python $R/tools/rocpd_rows.py $O/kt_results.db $O/shapes.json > $O/rows.txt 2>&1
rm -f $O/kt_results.db
cd $R; python bench.py --config c10_b512 --dev-local-batch 64 --force-dist --steps 200 --warmup 5 --no-cpu-baseline --no-g-step 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('b64 ms/step', d['ms_per_step'], d['config']['launch'])"
