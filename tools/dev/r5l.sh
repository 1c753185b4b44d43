# round 5, call l: settle the non-temporal-store question on sg2_512: five alternations never / always / by-size
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5l; mkdir -p $O; cd $R
V=$R/contrad_amd/csrc/variants
run() { if [ "$3" = "-" ]; then unset CONTRAD_HIP_LIB; else export CONTRAD_HIP_LIB=$V/libcontrad_$3.so; fi
  r=$(timeout 300 python bench.py --config $1 --no-cpu-baseline --no-g-step 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])")
  echo "st $1 [$2] $r" | tee -a $O/ab.txt; }
for rep in 1 2 3 4 5; do run sg2_512 never stoff; run sg2_512 always stall; run sg2_512 by-size -; done
