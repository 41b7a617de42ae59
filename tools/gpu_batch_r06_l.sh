#!/bin/bash
# round 6, call L: Q pre-scaled through the rotation table (one rounding of the self-attention's Q instead of two): the accuracy test, the
# module / block / model_fn / loop goldens, head-parallel bit-identity, full-size forward parity, then the fuzzer on attention
O=gpurun_out/r06
mkdir -p $O
( time timeout 1500 python -m pytest tests/test_kernels_gpu.py tests/test_dit_gpu.py tests/test_sequence_parallel_gpu.py tests/test_fulldepth_gpu.py tests/test_pipeline_call_gpu.py tests/test_fp8.py tests/test_training_gpu.py -m gpu -q -x -s --durations=6 ) > $O/prescale_tests_l.log 2>&1
echo "rc=$?" >> $O/prescale_tests_l.log
grep -v "^$\|SiLU C=" $O/prescale_tests_l.log | tail -24 | cut -c1-300
