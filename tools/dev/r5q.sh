# round 5, call q: wave priority (s_setprio) in the lean conv loops -- tools/experiments/wave_priority.diff built three ways
# (-DLEAN_PRIO=1: static priority per resident slot; 2: raised during a K-tile's MFMAs; 3: raised from the barrier to the
# next K-tile's first MFMA) against the shipped library, same box, alternating
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5q; mkdir -p $O; cd $R
V=$R/contrad_amd/csrc/variants
run() { if [ "$2" = "-" ]; then unset CONTRAD_HIP_LIB; else export CONTRAD_HIP_LIB=$V/libcontrad_$2.so; fi
  r=$(timeout 300 python bench.py --config $1 --no-cpu-baseline --no-g-step $3 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])")
  echo "q $1 [$2] $r" | tee -a $O/ab.txt; }
for rep in 1 2; do for v in - prio1 prio2 prio3; do run c10_b512 $v; done; done
for v in - prio1 prio2 prio3; do run sg2_512 $v "--steps 16 --warmup 2"; done
run c10_b512 -
