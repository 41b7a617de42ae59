#!/bin/bash
# PMC passes over the VAE convolutions of one production tile (tools/conv_bench.py: 96-channel direct kernel, 192-channel implicit GEMM,
# direct upsample): run on the GPU box from the repo root.  SQ / LDS counters; `bash tools/pmc_conv.sh hbm` for FETCH_SIZE / WRITE_SIZE / TCC.
set -u
OUT=$PWD/gpurun_out/r03/pmc_conv
mkdir -p $OUT
export TMPDIR=/tmp
export CONV_T=${CONV_T:-80}
if [ "${1:-sq}" = "hbm" ]; then
  SETS=("FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum")
  tag=hbm
else
  SETS=("SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MISC SQ_INSTS_VALU_MFMA_MOPS_BF16")
  tag=sq
fi
i=0
for SET in "${SETS[@]}"; do
  i=$((i+1))
  rocprofv3 --pmc $SET --kernel-trace -d $OUT/$tag$i -o a -- python3 tools/conv_bench.py > $OUT/$tag$i.log 2>&1
  python3 tools/rocpd_pmc.py $OUT/$tag$i/a_results.db "conv3d_c96|conv2d_up|gemm_ph" > $OUT/conv_${tag}$i.md 2>> $OUT/err.log
done
find $OUT -name "*.db" -delete
cat $OUT/conv_${tag}*.md | cut -c1-220
