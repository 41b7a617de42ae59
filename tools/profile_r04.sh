#!/bin/bash
# Round-4 counter collection (run on the GPU box from the repo root): tables under gpurun_out/r04/pmc/, then
# `python3 tools/pmc_static.py gpurun_out/r04/pmc profiles/r04/pmc` writes profiles/pmc_static.json (bench.py's static traffic).
# rocprofv3 rules of this pool: counters (--pmc) in their own passes with --kernel-trace only; the program directly after `--`.
set -u
OUT=$PWD/gpurun_out/r04/pmc
mkdir -p $OUT
export TMPDIR=/tmp
rocprofv3 -L > $OUT/counters_list.txt 2>&1
pass() {   # pass <tag> <counters> <kernel filter> -- <microbench args>
  local tag=$1 ctr=$2 filt=$3; shift 4
  rocprofv3 --pmc $ctr --kernel-trace -d $OUT/raw_$tag -o a -- python3 tools/microbench.py "$@" > $OUT/$tag.log 2>&1
  python3 tools/rocpd_pmc.py $OUT/raw_$tag/a_results.db "$filt" > $OUT/$tag.md 2>> $OUT/err.log
  rm -rf $OUT/raw_$tag
}
SQ1="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE"
SQ2="SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE"
SQ3="SQ_INSTS_MFMA SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_SALU SQ_INSTS_VMEM SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE"
# ---- self-attention (flash_attn_fwd_kernel3<2>, S = 32760 x 40 heads): traffic + where the non-MFMA cycles go
pass attn_FETCH_SIZE FETCH_SIZE "flash_attn|transpose_v32" -- attn --iters 2
pass attn_WRITE_SIZE WRITE_SIZE "flash_attn|transpose_v32" -- attn --iters 2
pass attn_SQ "$SQ1" "flash_attn" -- attn --iters 2
pass attn_SQ2 "$SQ2" "flash_attn" -- attn --iters 2
pass attn_SQ3 "$SQ3" "flash_attn" -- attn --iters 2
[ "${ONLY:-}" = attn ] && { ls -la $OUT; exit 0; }     # ONLY=attn: the attention passes alone (after a change to gf_attention.hip)
# ---- the block GEMMs, one shape per pass (so that a table row is one shape), bf16 and e4m3
for sh in dd ffn1 ffn2; do
  for c in FETCH_SIZE WRITE_SIZE; do
    pass gemm_${sh}_$c $c "gemm_a4" -- gemm --only $sh --iters 2
    pass gemm_fp8_${sh}_$c $c "gemm_a4" -- gemm --only $sh --fp8 --iters 2
  done
  pass gemm_${sh}_SQ "$SQ1" "gemm_a4" -- gemm --only $sh --iters 2
done
# ---- item 5 of the round-3 verdict: fabric bytes vs tile-group shape (row tiles x column tiles of the 32 workgroups an XCD runs at a time)
for g in 2 4 16; do
  export GF_A4_GROUP_M=$g
  pass gemm_ffn1_group${g}_FETCH_SIZE FETCH_SIZE "gemm_a4" -- gemm --only ffn1 --iters 2
  pass gemm_ffn2_group${g}_FETCH_SIZE FETCH_SIZE "gemm_a4" -- gemm --only ffn2 --iters 2
  unset GF_A4_GROUP_M
done
ls -la $OUT
