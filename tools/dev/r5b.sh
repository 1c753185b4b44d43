# round 5, call b: parity of the fused generator tail + where the ATen launches of a step come from + same-box A/B of the tail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5b; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_stylegan2_gpu.py tests/test_stylegan2_512_gpu.py tests/test_graph_gpu.py -q -m gpu -x -k "generator or modconv or graph_replay or fir4" > $O/pytest.log 2>&1; tail -4 $O/pytest.log
for c in sg2_32 sg2_512 c10_b512; do timeout 200 python tools/dev/aten_sources.py $c > $O/aten_$c.txt 2>&1; done
for rep in 1 2; do for f in 0 1; do for c in sg2_512 sg2_32; do
  CONTRAD_DEV_G_FUSE=$f timeout 200 python bench.py --config $c --no-cpu-baseline --no-g-step 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$c fuse=$f', d['ms_per_step'], d['value'])" | tee -a $O/ab.txt
done; done; done
