# round 5, call d: parity of the one-launch demodulation, the merged [reals | fakes] discriminator call, QPAD 8 as default;
# same-box A/Bs of both
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5d; mkdir -p $O; cd $R
timeout 900 python -m pytest tests -q -m gpu -x > $O/pytest.log 2>&1; tail -3 $O/pytest.log
ab() { local name=$1 cfg=$2 extra=$3; shift 3
  for rep in 1 2; do for E in "$@"; do
    [ "$E" = "-" ] && E=""
    r=$(env $E timeout 300 python bench.py --config $cfg $extra --no-cpu-baseline --no-g-step 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])")
    echo "$name $cfg rep$rep [$E] $r" | tee -a $O/ab.txt
  done; done
}
ab merged sg2_512 "" "CONTRAD_DEV_MERGED=0" "-"
ab legacy sg2_512 "" "CONTRAD_DEV_G_PREP=legacy" "-"
ab legacy sg2_32 "" "CONTRAD_DEV_G_PREP=legacy" "-"
for c in sg2_32 sg2_512; do timeout 200 python tools/dev/aten_sources.py $c > $O/aten_$c.txt 2>&1; done
