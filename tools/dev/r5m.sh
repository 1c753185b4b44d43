# round 5, call m: five alternations each -- headline: plain vs non-temporal activation stores; sg2_512: default vs FIR
# non-temporal stores (UF_NT=2) vs non-temporal epilogue operand loads (LEAN_LD_NT=1)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5m; mkdir -p $O; cd $R
V=$R/contrad_amd/csrc/variants
run() { if [ "$3" = "-" ]; then unset CONTRAD_HIP_LIB; else export CONTRAD_HIP_LIB=$V/libcontrad_$3.so; fi
  r=$(timeout 300 python bench.py --config $1 --no-cpu-baseline --no-g-step 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])")
  echo "m $1 [$2] $r" | tee -a $O/ab.txt; }
for rep in 1 2 3 4 5; do run c10_b512 plain stoff; run c10_b512 default -; done
for rep in 1 2 3 4 5; do run sg2_512 default -; run sg2_512 fir-nt-stores ufnt2; run sg2_512 epi-nt-loads ldnt; done
for rep in 1 2 3; do run sg2_32 plain stoff; run sg2_32 default -; done
