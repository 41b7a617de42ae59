#!/bin/bash
# Round-3 profile collection (run on the GPU box from the repo root): writes everything under gpurun_out/r03/prof/.
# rocprofv3 rules of this pool: counters (--pmc) in their own passes with --kernel-trace only; the program directly after `--`.
set -u
OUT=$PWD/gpurun_out/r03/prof
mkdir -p $OUT
export TMPDIR=/tmp
SQ="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE"
# 1. per-kernel time of one denoise step + VAE decode, bf16 (the same command as bench.py --steps 1) and fp8 (config 5)
rocprofv3 --kernel-trace --stats -d $OUT/trace_bf16 -o bench -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline > $OUT/bench_steps1_bf16.json.log 2> $OUT/trace_bf16.err
python3 tools/rocpd_table.py $OUT/trace_bf16/bench_results.db 40 > $OUT/bench_steps1_bf16_by_kernel_and_grid.md 2>> $OUT/trace_bf16.err
rocprofv3 --kernel-trace --stats -d $OUT/trace_fp8 -o bench -- python3 bench.py --fp8 --steps 1 --warmup 0 --no-cpu-baseline > $OUT/bench_steps1_fp8.json.log 2> $OUT/trace_fp8.err
python3 tools/rocpd_table.py $OUT/trace_fp8/bench_results.db 40 > $OUT/bench_steps1_fp8_by_kernel_and_grid.md 2>> $OUT/trace_fp8.err
# 2. HBM traffic + matrix-pipe busy of the self-attention launch (flash_attn_fwd_kernel3) of THIS build: bench.py's roofline.traffic
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace -d $OUT/pmc_attn_$c -o a -- python3 tools/microbench.py attn --iters 2 > $OUT/pmc_attn_$c.log 2>&1
  python3 tools/rocpd_pmc.py $OUT/pmc_attn_$c/a_results.db "flash_attn|transpose_v32" > $OUT/attn_$c.md 2>> $OUT/err.log
done
rocprofv3 --pmc $SQ --kernel-trace -d $OUT/pmc_attn_SQ -o a -- python3 tools/microbench.py attn --iters 2 > $OUT/pmc_attn_SQ.log 2>&1
python3 tools/rocpd_pmc.py $OUT/pmc_attn_SQ/a_results.db "flash_attn" > $OUT/attn_SQ.md 2>> $OUT/err.log
# 3. the 4-wave GEMMs, bf16 and fp8 (tools/microbench.py gemm --fp8 runs both on the three block shapes): matrix-pipe busy, L2, HBM
rocprofv3 --pmc $SQ --kernel-trace -d $OUT/pmc_gemm_SQ -o a -- python3 tools/microbench.py gemm --fp8 --iters 2 > $OUT/pmc_gemm_SQ.log 2>&1
python3 tools/rocpd_pmc.py $OUT/pmc_gemm_SQ/a_results.db "gemm_a4" > $OUT/gemm_SQ.md 2>> $OUT/err.log
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_BUSY_sum --kernel-trace -d $OUT/pmc_gemm_TCC -o a -- python3 tools/microbench.py gemm --fp8 --iters 2 > $OUT/pmc_gemm_TCC.log 2>&1
python3 tools/rocpd_pmc.py $OUT/pmc_gemm_TCC/a_results.db "gemm_a4" > $OUT/gemm_TCC.md 2>> $OUT/err.log
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace -d $OUT/pmc_gemm_$c -o a -- python3 tools/microbench.py gemm --fp8 --iters 2 > $OUT/pmc_gemm_$c.log 2>&1
  python3 tools/rocpd_pmc.py $OUT/pmc_gemm_$c/a_results.db "gemm_a4" > $OUT/gemm_$c.md 2>> $OUT/err.log
done
# the databases are large: only the tables travel back
find $OUT -name "*.db" -delete
find $OUT -name "*.csv" -size +1M -delete
ls -la $OUT
