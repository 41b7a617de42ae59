#!/bin/bash
# round 5, call B: the padded-layout convolution kernel — parity tests first, then the A/B against the implicit GEMM, then the VAE
mkdir -p gpurun_out/r05
( timeout 900 python -m pytest tests/test_vae.py -m gpu -x -q -k "padded or both_conv_paths or tiled or golden" ) > gpurun_out/r05/conv_a4_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r05/conv_a4_tests.log
tail -15 gpurun_out/r05/conv_a4_tests.log
( timeout 600 python tools/conv_a4_bench.py ) > gpurun_out/r05/conv_a4_bench.log 2>&1
cat gpurun_out/r05/conv_a4_bench.log | grep -v "^MIOpen\|amdgpu.ids"
( timeout 300 python tools/vae_profile.py decode 3; timeout 300 python tools/vae_profile.py encode 3 ) > gpurun_out/r05/vae_conv_a4.log 2>&1
grep -v "^MIOpen\|amdgpu.ids" gpurun_out/r05/vae_conv_a4.log
