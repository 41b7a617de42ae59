#!/bin/bash
# Round-3 final measurements on the shipped build (GPU box, repo root): default bench lines (bf16, fp8), the per-kernel tables of
# one step under rocprofv3 --kernel-trace --stats, the 50-step un-extrapolated runs.  Output: gpurun_out/r03/final/.
set -u
OUT=$PWD/gpurun_out/r03/final
mkdir -p $OUT
export TMPDIR=/tmp
python3 bench.py > $OUT/bench_bf16_final.json.log 2> $OUT/bench_bf16_final.err
python3 bench.py --fp8 > $OUT/bench_fp8_final.json.log 2> $OUT/bench_fp8_final.err
rocprofv3 --kernel-trace --stats -d $OUT/trace_bf16 -o bench -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline > $OUT/bench_steps1_bf16_final.json.log 2> $OUT/trace_bf16.err
python3 tools/rocpd_table.py $OUT/trace_bf16/bench_results.db 40 > $OUT/bench_steps1_bf16_final_by_kernel_and_grid.md 2>> $OUT/trace_bf16.err
rocprofv3 --kernel-trace --stats -d $OUT/trace_fp8 -o bench -- python3 bench.py --fp8 --steps 1 --warmup 0 --no-cpu-baseline > $OUT/bench_steps1_fp8_final.json.log 2> $OUT/trace_fp8.err
python3 tools/rocpd_table.py $OUT/trace_fp8/bench_results.db 40 > $OUT/bench_steps1_fp8_final_by_kernel_and_grid.md 2>> $OUT/trace_fp8.err
find $OUT -name "*.db" -delete
find $OUT -name "*.csv" -size +1M -delete
python3 bench.py --steps 50 --warmup 1 --no-cpu-baseline > $OUT/bench_bf16_50steps_final.json.log 2>> $OUT/bench_bf16_final.err
python3 bench.py --fp8 --steps 50 --warmup 1 --no-cpu-baseline > $OUT/bench_fp8_50steps_final.json.log 2>> $OUT/bench_fp8_final.err
ls -la $OUT
