#!/bin/bash
# round 6, call R: attention forward + backward vs fp64 at peaky logits (plain vs pre-scaled q, self- and cross-attention shape); the training tests
O=gpurun_out/r06
mkdir -p $O
timeout 300 python tools/attnbwd_precision.py > $O/attnbwd_precision.log 2>&1; timeout 300 python tools/attnbwd_precision.py --skv 512 >> $O/attnbwd_precision.log 2>&1
grep -v amdgpu.ids $O/attnbwd_precision.log | cut -c1-200
( time timeout 900 python -m pytest tests/test_training_gpu.py tests/test_kernels_gpu.py -m gpu -q -x --durations=5 ) > $O/train_tests_r.log 2>&1
echo "rc=$?" >> $O/train_tests_r.log; tail -14 $O/train_tests_r.log | cut -c1-300
