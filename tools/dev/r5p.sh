# round 5, call p: the non-temporal store form hoisted out of the epilogue's store loops and confined to the narrow
# instances: full GPU suite (incl. the nt-vs-plain equivalence tests), alternations never / default on headline + sg2_512
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5p; mkdir -p $O; cd $R
V=$R/contrad_amd/csrc/variants
timeout 1000 python -m pytest tests -q -m gpu -x > $O/pytest.log 2>&1; grep -E "passed|failed" $O/pytest.log
run() { if [ "$3" = "-" ]; then unset CONTRAD_HIP_LIB; else export CONTRAD_HIP_LIB=$V/libcontrad_$3.so; fi
  r=$(timeout 300 python bench.py --config $1 --no-cpu-baseline --no-g-step 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])")
  echo "p $1 [$2] $r" | tee -a $O/ab.txt; }
for rep in 1 2 3 4 5; do run c10_b512 never stoff; run c10_b512 default -; done
for rep in 1 2 3 4; do run sg2_512 never stoff; run sg2_512 default -; done
