#!/bin/bash
# round 5, call G (shipped build): smoke(), the rocprofv3 kernel trace of the bench command (per-kernel table the roofline figure must agree with),
# the un-extrapolated 50-step loops in bf16 and fp8, the training step
O=gpurun_out/r05
mkdir -p $O
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke_shipped.log 2>&1; tail -2 $O/smoke_shipped.log
rocprofv3 --kernel-trace --stats -d $O/trace_shipped -o bench -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --config5-steps 0 --peaky-steps 0 --no-preloop > $O/bench_steps5_shipped.json.log 2> $O/trace_shipped.err
python3 tools/rocpd_table.py $O/trace_shipped/bench_results.db 40 > $O/bench_steps5_shipped_by_kernel_and_grid.md 2>> $O/trace_shipped.err
head -12 $O/bench_steps5_shipped_by_kernel_and_grid.md | cut -c1-200
rm -rf $O/trace_shipped
( timeout 900 python bench.py --steps 50 --warmup 2 --no-cpu-baseline --config5-steps 0 --peaky-steps 0 --no-preloop ) > $O/bench_bf16_50steps.json.log 2> $O/bench_bf16_50steps.err; tail -c 400 $O/bench_bf16_50steps.json.log | head -c 300; echo
( timeout 900 python bench.py --fp8 --steps 50 --warmup 2 --no-cpu-baseline --no-preloop ) > $O/bench_fp8_50steps.json.log 2> $O/bench_fp8_50steps.err; echo "fp8 rc=$?"
( timeout 900 python tools/train_bench.py --layers 40 --cn-layers 10 --steps 2 ) > $O/train_step_a14b.log 2>&1; tail -3 $O/train_step_a14b.log | cut -c1-300
