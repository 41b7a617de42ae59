#!/bin/bash
# The product sources carry only what ships.  The kernels and diagnostic builds that were measured and dropped (attention kernel 1,
# the ORMAX / head-major / what-if / clock / stamp variants of kernel 3, the `sl` / `sl8` / bf16-v1 GEMMs, the first attention
# backward kernels, the 32-key dK/dV kernel, the pre-scaled-Q backward, the -DCD_WHATIF / -DCD_SHARE_A=0 builds of the 96-channel direct
# convolution, the scalar-pair LayerNorm build, the env-variable option channel that selected them) live as
# patches under tools/patches/.  This script re-creates the experimental source tree under build/experimental/ (csrc + include);
# the A/B tools (tools/attn_ab.py, attn_clock.py, attn_whatif.py, attn_stamps.py, attnbwd_ab.py, gemm_*stamps.py, gemm_*whatif.py,
# gemm_variants.py) build from it (GF_CSRC overrides the location).  Nothing built from it is ever loaded by goal_force_amd.
set -e
cd "$(dirname "$0")/.."
rm -rf build/experimental
mkdir -p build/experimental
cp -r goal_force_amd/csrc build/experimental/csrc
cp -r include build/experimental/include
( cd build/experimental/csrc
  patch -s -p0 gf_attention.hip < ../../../tools/patches/attention_experiments.patch
  patch -s -p0 gf_attention_bwd.hip < ../../../tools/patches/attention_bwd_experiments.patch
  patch -s -p0 gf_gemm.hip < ../../../tools/patches/gemm_experiments.patch
  patch -s -p0 gf_conv_direct.hip < ../../../tools/patches/conv_direct_experiments.patch
  patch -s -p0 gf_rowops.hip < ../../../tools/patches/rowops_experiments.patch
  patch -s -p0 < ../../../tools/patches/abi_experiments.patch )
( cd build/experimental && patch -s -p0 include/goalforce.h < ../../tools/patches/header_experiments.patch )
A4_WHATIF_VARIANTS=1 A4_OUT=build/experimental/csrc/gf_gemm_a4_loop.inc python3 tools/gen_gemm_a4.py > /dev/null      # + the what-if loops
echo "build/experimental/{csrc,include}: the experimental tree (kernel 1, sl GEMMs, v1 backward, GF_K3_* / GF_*_WHATIF / *_STAMP builds)"
