#!/bin/bash
# Re-creates tools/patches/*_experiments.patch: each is `diff -u <product file> <the same file in the round-4 tree (commit 84a39fb)>`, i.e.
# applying it to the product file gives back the round-4 file with every experimental kernel / diagnostic branch / env selector in it.
# Run after any edit to one of these product files (needs the git history; the GPU box never runs this).
set -e
cd "$(dirname "$0")/../.."
R4=84a39fb
mk() {   # mk <patch name> <file>...
  local out=tools/patches/$1; shift
  : > $out
  for f in "$@"; do
    git show $R4:$f > /tmp/_r4_file
    ( cd $(dirname $f) && diff -u $(basename $f) /tmp/_r4_file | sed "2s#/tmp/_r4_file#$(basename $f).r4#" ) >> $out || true
  done
}
mk attention_experiments.patch goal_force_amd/csrc/gf_attention.hip
mk attention_bwd_experiments.patch goal_force_amd/csrc/gf_attention_bwd.hip
mk gemm_experiments.patch goal_force_amd/csrc/gf_gemm.hip
mk abi_experiments.patch goal_force_amd/csrc/gf_abi.hip goal_force_amd/csrc/gf_common.h
mk header_experiments.patch include/goalforce.h
mk conv_direct_experiments.patch goal_force_amd/csrc/gf_conv_direct.hip
mk rowops_experiments.patch goal_force_amd/csrc/gf_rowops.hip
wc -l tools/patches/*_experiments.patch
