#!/bin/bash
# Round-2 profile collection (run on the GPU box from the repo root): writes everything under gpurun_out/r02/.
# rocprofv3 rules of this pool: counters (--pmc) in their own passes with --kernel-trace only; the program directly after `--`.
set -u
OUT=$PWD/gpurun_out/r02
mkdir -p $OUT
export TMPDIR=/tmp
# 1. per-kernel time of one denoise step + VAE decode (the same command as bench.py --steps 1)
rocprofv3 --kernel-trace --stats -d $OUT/trace -o bench -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline > $OUT/bench_steps1.json.log 2> $OUT/trace.err
python3 tools/rocpd_table.py $OUT/trace/bench_results.db 40 > $OUT/bench_steps1_by_kernel_and_grid.md 2>> $OUT/trace.err
# 2. HBM traffic + matrix-pipe busy of the self-attention launch (transpose_v32 + flash_attn_fwd_kernel3<true>)
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace -d $OUT/pmc_attn_$c -o a -- python3 tools/microbench.py attn --iters 2 > $OUT/pmc_attn_$c.log 2>&1
done
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --kernel-trace -d $OUT/pmc_attn_SQ -o a -- python3 tools/microbench.py attn --iters 2 > $OUT/pmc_attn_SQ.log 2>&1
# 3. the 4-wave GEMM: matrix-pipe busy, L2 hit rate / busy, HBM traffic
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --kernel-trace -d $OUT/pmc_gemm_SQ -o a -- python3 tools/microbench.py gemm --iters 2 > $OUT/pmc_gemm_SQ.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_BUSY_sum --kernel-trace -d $OUT/pmc_gemm_TCC -o a -- python3 tools/microbench.py gemm --iters 2 > $OUT/pmc_gemm_TCC.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace -d $OUT/pmc_gemm_$c -o a -- python3 tools/microbench.py gemm --iters 2 > $OUT/pmc_gemm_$c.log 2>&1
done
find $OUT -name "*.csv" | head -40
ls -la $OUT
