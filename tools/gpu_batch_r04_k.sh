#!/bin/bash
# Round 4, batch K (GPU box, repo root): the whole GPU suite and the driver's bench command on the final tree.
O=gpurun_out/r04
mkdir -p $O
python -m pytest tests -q -m gpu > $O/gpu_suite_final3.log 2>&1; tail -4 $O/gpu_suite_final3.log
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd_final3.json.log 2> $O/bench_driver_cmd_final3.err; tail -c 300 $O/bench_driver_cmd_final3.json.log; tail -2 $O/bench_driver_cmd_final3.err
