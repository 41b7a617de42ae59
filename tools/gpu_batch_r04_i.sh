#!/bin/bash
# Round 4, batch I (GPU box, repo root): how much of a denoising step the GPU sits idle between kernels (kernel trace -> tools/rocpd_gaps.py).
O=gpurun_out/r04
mkdir -p $O
export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $O/trace_gaps -o bench -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --config5-steps 0 > $O/bench_gaps.json.log 2> $O/trace_gaps.err
python3 tools/rocpd_gaps.py $O/trace_gaps/bench_results.db 120 360 > $O/step_idle_gaps.md 2>> $O/trace_gaps.err
cat $O/step_idle_gaps.md | cut -c1-200
rm -rf $O/trace_gaps
