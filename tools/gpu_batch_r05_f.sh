#!/bin/bash
# round 5, call F: schedule variants of the padded-layout convolution's K loop (tools/conv_a4_variants.sh), two interleaved rounds; then the new edge-shape tests
O=gpurun_out/r05
mkdir -p $O
: > $O/conv_a4_variants.log
for rnd in 1 2; do
  for d in build/variants/conv_*; do
    echo "== $d (round $rnd)" >> $O/conv_a4_variants.log
    GOALFORCE_HIP_LIB=$PWD/$d/libgoalforce_hip.so timeout 300 python tools/conv_a4_bench.py 2>&1 | grep -v "^MIOpen\|amdgpu.ids" | grep "padded\|identical" >> $O/conv_a4_variants.log
  done
done
grep "==\|192->192 81\|384->384 41\|False" $O/conv_a4_variants.log
( timeout 600 python -m pytest tests/test_vae.py -m gpu -x -q -k "padded" ) > $O/conv_a4_edge_tests.log 2>&1; tail -3 $O/conv_a4_edge_tests.log
