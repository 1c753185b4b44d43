# round 5, call c: full GPU suite on the new generator tables / SN prefetch, then same-box A/Bs:
#   generator table prep (legacy per-layer ATen vs one-launch tables), SN prefetch on / off (batch 512 and per-rank 64),
#   quad-layout LDS padding 16 vs 8 (bank conflicts of ds_write_b128) incl. the conflict counters
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5c; mkdir -p $O; cd $R
timeout 900 python -m pytest tests -q -m gpu -x > $O/pytest.log 2>&1; tail -4 $O/pytest.log
ab() { # name config extra-args env...
  local name=$1 cfg=$2 extra=$3; shift 3
  for rep in 1 2; do for E in "$@"; do
    [ "$E" = "-" ] && E=""
    r=$(env $E timeout 300 python bench.py --config $cfg $extra --no-cpu-baseline --no-g-step 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])")
    echo "$name $cfg rep$rep [$E] $r" | tee -a $O/ab.txt
  done; done
}
ab gprep sg2_512 "" "CONTRAD_DEV_G_PREP=legacy" "-"
ab gprep sg2_32 "" "CONTRAD_DEV_G_PREP=legacy" "-"
ab snpre c10_b512 "" "CONTRAD_DEV_SN_PREFETCH=0" "-"
ab snpre c10_b512 "--dev-local-batch 64 --force-dist --steps 200 --warmup 5" "CONTRAD_DEV_SN_PREFETCH=0" "-"
Q=$R/contrad_amd/csrc/variants/libcontrad_qpad8.so
ab qpad c10_b512 "" "-" "CONTRAD_HIP_LIB=$Q"
ab qpad sg2_512 "" "-" "CONTRAD_HIP_LIB=$Q"
ab qpad sg2_32 "" "-" "CONTRAD_HIP_LIB=$Q"
CONV_LAYERS=3 bash tools/pmc_lds.sh base qpad8 > $O/pmc_lds.txt 2>&1; tail -30 $O/pmc_lds.txt
