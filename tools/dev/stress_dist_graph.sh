#!/bin/bash
# dev: the graph-with-collectives worker repeatedly (an intermittent ProcessGroupNCCL watchdog abort was seen once)
cd $GRAFT_REPO_ROOT
for i in $(seq 1 ${1:-6}); do
  port=$((29600 + i))
  timeout 200 python tests/dist_graph_worker.py $port > gpurun_out/stress_$i.out 2> gpurun_out/stress_$i.err
  echo "run $i rc=$? : $(grep -c '^OK' gpurun_out/stress_$i.out) scenarios OK; $(grep -m1 -o 'watchdog thread terminated[^:]*: [^.]*' gpurun_out/stress_$i.err)"
done
