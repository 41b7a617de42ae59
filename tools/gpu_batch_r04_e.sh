#!/bin/bash
# Round 4, final GPU batch (GPU box, repo root): whole suite, counters of the final kernels -> profiles/pmc_static.json, the driver's bench
# command with the traffic figures in it, whole-loop benches, kernel trace, the 4-step full-size parity on the final build.
O=gpurun_out/r04
mkdir -p $O
export TMPDIR=/tmp
python -m pytest tests -q -m gpu > $O/gpu_suite_final.log 2>&1; tail -4 $O/gpu_suite_final.log
bash tools/profile_r04.sh > $O/profile_r04.log 2>&1; tail -2 $O/profile_r04.log
python tools/pmc_static.py $O/pmc profiles/r04/pmc > $O/pmc_static.log 2>&1; cat $O/pmc_static.log; cp profiles/pmc_static.json $O/pmc_static.json
python tools/attn_ab.py --rounds 4 > $O/attn_ab_final.log 2>&1; cat $O/attn_ab_final.log
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd.json.log 2> $O/bench_driver_cmd.err; tail -c 400 $O/bench_driver_cmd.json.log; tail -3 $O/bench_driver_cmd.err
python bench.py --steps 50 --warmup 1 --no-cpu-baseline --config5-steps 0 > $O/bench_bf16_50steps.json.log 2>&1
python bench.py --fp8 --steps 50 --warmup 1 --no-cpu-baseline > $O/bench_fp8_50steps.json.log 2>&1
rocprofv3 --kernel-trace --stats -d $O/trace_bf16 -o bench -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --config5-steps 0 > $O/bench_steps1_bf16.json.log 2> $O/trace_bf16.err
python3 tools/rocpd_table.py $O/trace_bf16/bench_results.db 40 > $O/bench_steps1_bf16_by_kernel_and_grid.md 2>> $O/trace_bf16.err
find $O/trace_bf16 -name "*.db" -delete; find $O/trace_bf16 -name "*.csv" -size +1M -delete
timeout 1200 python tests/fullsize_parity.py --steps 4 --fp8 --out $O/fullsize_parity_4step_final.json > $O/fullsize_parity_4step_final.log 2>&1; tail -8 $O/fullsize_parity_4step_final.log | cut -c1-400
