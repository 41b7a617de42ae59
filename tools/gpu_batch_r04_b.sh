#!/bin/bash
# Round 4, GPU batch B (GPU box, repo root): counters of the final kernels, the driver's bench command, whole-loop benches, kernel trace.
O=gpurun_out/r04
mkdir -p $O
export TMPDIR=/tmp
bash tools/profile_r04.sh > $O/profile_r04.log 2>&1; tail -3 $O/profile_r04.log
python tools/gemm_groupm.py 8 4 2 16 > $O/gemm_groupm.log 2>&1; cat $O/gemm_groupm.log
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd.json.log 2> $O/bench_driver_cmd.err; tail -c 600 $O/bench_driver_cmd.json.log
python bench.py --steps 50 --warmup 1 --no-cpu-baseline --config5-steps 0 > $O/bench_bf16_50steps.json.log 2>&1
python bench.py --fp8 --steps 50 --warmup 1 --no-cpu-baseline > $O/bench_fp8_50steps.json.log 2>&1
rocprofv3 --kernel-trace --stats -d $O/trace_bf16 -o bench -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --config5-steps 0 > $O/bench_steps1_bf16.json.log 2> $O/trace_bf16.err
python3 tools/rocpd_table.py $O/trace_bf16/bench_results.db 40 > $O/bench_steps1_bf16_by_kernel_and_grid.md 2>> $O/trace_bf16.err
cp $O/trace_bf16/bench_kernel_stats.csv $O/bench_steps1_bf16_kernel_stats.csv 2>/dev/null
find $O/trace_bf16 -name "*.db" -delete; find $O/trace_bf16 -name "*.csv" -size +1M -delete
python tools/train_bench.py > $O/train_step_a14b.log 2>&1; tail -3 $O/train_step_a14b.log
