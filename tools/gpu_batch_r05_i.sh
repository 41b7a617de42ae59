#!/bin/bash
# round 5, call I: the head-parallel bench test (new), then the whole suite once more on the final tree
O=gpurun_out/r05
mkdir -p $O
( time timeout 1200 python -m pytest tests/test_bench_gpu.py -m gpu -x -q -k "head_parallel" ) > $O/bench_sp_test.log 2>&1; tail -5 $O/bench_sp_test.log | cut -c1-300
( time timeout 1500 python -m pytest tests -m gpu -x -q --durations=6 ) > $O/gpu_suite_final.log 2>&1
echo "suite rc=$?" >> $O/gpu_suite_final.log; tail -12 $O/gpu_suite_final.log
