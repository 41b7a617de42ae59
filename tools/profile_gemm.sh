#!/bin/bash
# Counter passes over the 4-wave GEMM (tools/microbench.py gemm --ref: our kernel and hipBLASLt on the same shapes).  GPU box, repo root.
set -u
OUT=$PWD/gpurun_out/r02c
mkdir -p $OUT
export TMPDIR=/tmp
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --kernel-trace -d $OUT/pmc_gemm_SQ -o a -- python3 tools/microbench.py gemm --ref --iters 2 > $OUT/pmc_gemm_SQ.log 2>&1
python3 tools/rocpd_pmc.py $OUT/pmc_gemm_SQ/a_results.db "gemm_a4|Cijk" > $OUT/gemm_SQ.md 2>> $OUT/err.log
find $OUT -name "*.db" -size +20M -delete
cat $OUT/gemm_SQ.md
grep -v amdgpu.ids $OUT/pmc_gemm_SQ.log | tail -12
