#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -5
timeout 300 python bench.py --config sg2_32 --no-cpu-baseline --no-g-step 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('sg2_32', d['ms_per_step'], d['value'])"
timeout 300 python bench.py --config c10_b512 --no-cpu-baseline --no-g-step 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c10', d['ms_per_step'], d['value'])"
