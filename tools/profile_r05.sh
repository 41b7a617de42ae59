#!/bin/bash
# Round-5 counter collection (run on the GPU box from the repo root): tables under gpurun_out/r05/pmc/, then
# `python3 tools/pmc_static.py gpurun_out/r05/pmc profiles/r05/pmc` writes profiles/pmc_static.json (bench.py's static traffic).
# rocprofv3 rules of this pool: counters (--pmc) in their own passes with --kernel-trace only; the program directly after `--`.
set -u
OUT=$PWD/gpurun_out/r05/pmc
mkdir -p $OUT
export TMPDIR=/tmp
pass() {   # pass <tag> <counters> <kernel filter> <program> <args...>
  local tag=$1 ctr=$2 filt=$3 prog=$4; shift 4
  rocprofv3 --pmc $ctr --kernel-trace -d $OUT/raw_$tag -o a -- python3 $prog "$@" > $OUT/$tag.log 2>&1
  python3 tools/rocpd_pmc.py $OUT/raw_$tag/a_results.db "$filt" > $OUT/$tag.md 2>> $OUT/err.log
  rm -rf $OUT/raw_$tag
}
SQ1="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE"
# ---- the VAE's dominant convolution launches (BASELINE config 4): one shape per pass
for sh in l2 l1 c96; do
  pass vae_${sh}_FETCH_SIZE FETCH_SIZE "conv" tools/conv_pmc.py $sh
  pass vae_${sh}_WRITE_SIZE WRITE_SIZE "conv" tools/conv_pmc.py $sh
  pass vae_${sh}_SQ "$SQ1" "conv" tools/conv_pmc.py $sh
done
[ "${ONLY:-}" = vae ] && { ls -la $OUT; exit 0; }
# ---- self-attention (flash_attn_fwd_kernel3<2>, S = 32760 x 40 heads)
pass attn_FETCH_SIZE FETCH_SIZE "flash_attn|transpose_v32" tools/microbench.py attn --iters 2
pass attn_WRITE_SIZE WRITE_SIZE "flash_attn|transpose_v32" tools/microbench.py attn --iters 2
pass attn_SQ "$SQ1" "flash_attn" tools/microbench.py attn --iters 2
# ---- the block GEMMs, one shape per pass, bf16 and e4m3
for sh in dd ffn1 ffn2; do
  for c in FETCH_SIZE WRITE_SIZE; do
    pass gemm_${sh}_$c $c "gemm_a4" tools/microbench.py gemm --only $sh --iters 2
    pass gemm_fp8_${sh}_$c $c "gemm_a4" tools/microbench.py gemm --only $sh --fp8 --iters 2
  done
done
ls -la $OUT
