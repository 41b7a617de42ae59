#!/bin/bash
# Round 4, batch H (GPU box, repo root): smoke() and the rocprofv3 kernel trace of bench.py on the shipped build.
O=gpurun_out/r04
mkdir -p $O
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke_shipped.log 2>&1; tail -3 $O/smoke_shipped.log
rocprofv3 --kernel-trace --stats -d $O/trace_shipped -o bench -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --config5-steps 0 > $O/bench_steps5_shipped.json.log 2> $O/trace_shipped.err
python3 tools/rocpd_table.py $O/trace_shipped/bench_results.db 40 > $O/bench_steps5_shipped_by_kernel_and_grid.md 2>> $O/trace_shipped.err
head -12 $O/bench_steps5_shipped_by_kernel_and_grid.md | cut -c1-200
find $O/trace_shipped -name "*.db" -delete; find $O/trace_shipped -name "*.csv" -size +1M -delete
tail -c 300 $O/bench_steps5_shipped.json.log
