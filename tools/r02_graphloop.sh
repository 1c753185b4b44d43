#!/bin/bash
# dev run: training loops with --graph against the eager loops; bench after the in-graph re-pack of G's weights
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_graph_gpu.py tests/test_train_loop_gpu.py -x -q > gpurun_out/graphloop_tests.log 2>&1
tail -15 gpurun_out/graphloop_tests.log
timeout 600 python bench.py --no-cpu-baseline > gpurun_out/graphloop_bench.json 2> gpurun_out/graphloop_bench.err
tail -c 3000 gpurun_out/graphloop_bench.json
