#!/bin/bash
# Attention-only counter passes (kernel 3) + a bench line + one traced step.  Run on the GPU box from the repo root.
set -u
OUT=$PWD/gpurun_out/r02b
mkdir -p $OUT
export TMPDIR=/tmp
python3 bench.py > $OUT/bench_default.json.log 2> $OUT/bench_default.err
rocprofv3 --kernel-trace --stats -d $OUT/trace -o bench -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline > $OUT/bench_steps1.json.log 2> $OUT/trace.err
python3 tools/rocpd_table.py $OUT/trace/bench_results.db 40 > $OUT/bench_steps1_by_kernel_and_grid.md 2>> $OUT/trace.err
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace -d $OUT/pmc_attn_$c -o a -- python3 tools/microbench.py attn --iters 2 > $OUT/pmc_attn_$c.log 2>&1
  python3 tools/rocpd_pmc.py $OUT/pmc_attn_$c/a_results.db > $OUT/attn_$c.md 2>> $OUT/trace.err
done
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --kernel-trace -d $OUT/pmc_attn_SQ -o a -- python3 tools/microbench.py attn --iters 2 > $OUT/pmc_attn_SQ.log 2>&1
python3 tools/rocpd_pmc.py $OUT/pmc_attn_SQ/a_results.db > $OUT/attn_SQ.md 2>> $OUT/trace.err
find $OUT -name "*.db" -size +20M -delete
ls -la $OUT; cat $OUT/bench_default.json.log; cat $OUT/attn_*.md
