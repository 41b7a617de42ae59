#!/bin/bash
# round 5, call A: the whole GPU suite on the cleaned-up library, then the driver's bench command
mkdir -p gpurun_out/r05
( time timeout 1500 python -m pytest tests -m gpu -x -q --durations=15 ) > gpurun_out/r05/gpu_suite_a.log 2>&1
echo "suite rc=$?" >> gpurun_out/r05/gpu_suite_a.log
( time timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 ) > gpurun_out/r05/bench_driver_cmd_a.json.log 2> gpurun_out/r05/bench_driver_cmd_a.stderr.log
echo "bench rc=$?" >> gpurun_out/r05/bench_driver_cmd_a.stderr.log
tail -3 gpurun_out/r05/gpu_suite_a.log; tail -c 1500 gpurun_out/r05/bench_driver_cmd_a.json.log
