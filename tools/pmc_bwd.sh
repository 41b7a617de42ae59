#!/bin/bash
# PMC passes over the flash-attention backward (tools/microbench.py attnbwd): run on the GPU box from the repo root.
#   bash tools/pmc_bwd.sh            SQ / LDS counters
#   bash tools/pmc_bwd.sh hbm        FETCH_SIZE / WRITE_SIZE (HBM traffic per launch)
set -u
OUT=$PWD/gpurun_out/r03/pmc_bwd
mkdir -p $OUT
export TMPDIR=/tmp
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
  rocprofv3 --pmc $SET --kernel-trace -d $OUT/$tag$i -o a -- python3 tools/microbench.py attnbwd --iters 2 > $OUT/$tag$i.log 2>&1
  python3 tools/rocpd_pmc.py $OUT/$tag$i/a_results.db "attn_bwd_d|flash_attn" > $OUT/bwd_${tag}$i.md 2>> $OUT/err.log
done
find $OUT -name "*.db" -delete
cat $OUT/bwd_${tag}*.md | cut -c1-200
