# round 5, call e: parity with packed weight gradients accumulated in the pack's slots (contrad_conv2d_wgrad accumulate),
# same-box A/B against autograd's adds
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5e; mkdir -p $O; cd $R
timeout 900 python -m pytest tests -q -m gpu -x > $O/pytest.log 2>&1; tail -3 $O/pytest.log
ab() { local name=$1 cfg=$2 extra=$3; shift 3
  for rep in 1 2 3; do for E in "$@"; do
    [ "$E" = "-" ] && E=""
    r=$(env $E timeout 300 python bench.py --config $cfg $extra --no-cpu-baseline --no-g-step 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])")
    echo "$name $cfg rep$rep [$E] $r" | tee -a $O/ab.txt
  done; done
}
ab slotacc sg2_32 "" "CONTRAD_DEV_SLOT_ACC=0" "-"
timeout 200 python tools/dev/aten_sources.py sg2_32 > $O/aten_sg2_32.txt 2>&1
