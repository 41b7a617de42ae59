set -u
OUT=$PWD/gpurun_out/r02
mkdir -p $OUT
export TMPDIR=/tmp
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_BUSY_sum --kernel-trace -d $OUT/pmc_ref_TCC -o a -- python3 tools/microbench.py gemm --ref --iters 2 > $OUT/pmc_ref_TCC.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --kernel-trace -d $OUT/pmc_ref_SQ -o a -- python3 tools/microbench.py gemm --ref --iters 2 > $OUT/pmc_ref_SQ.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/pmc_ref_FETCH -o a -- python3 tools/microbench.py gemm --ref --iters 2 > $OUT/pmc_ref_FETCH.log 2>&1
rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TA_TA_BUSY_sum TCP_TA_DATA_STALL_CYCLES_sum --kernel-trace -d $OUT/pmc_ref_TCP -o a -- python3 tools/microbench.py gemm --ref --iters 2 > $OUT/pmc_ref_TCP.log 2>&1
for d in TCC SQ FETCH TCP; do python3 tools/rocpd_pmc.py $OUT/pmc_ref_$d/a_results.db "Cijk|gemm_a4" > $OUT/ref_$d.md; done
cat $OUT/ref_TCC.md $OUT/ref_FETCH.md $OUT/ref_TCP.md | cut -c1-220
