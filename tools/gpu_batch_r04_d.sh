#!/bin/bash
# Round 4, GPU batch D: the pad-key folding of the cross-attention (tests, cross-attention timing, bench with / without it).
O=gpurun_out/r04
mkdir -p $O
python -m pytest tests/test_kernels_gpu.py tests/test_dit_gpu.py tests/test_canny.py tests/test_training_gpu.py tests/test_fulldepth_gpu.py tests/test_fp8.py tests/test_sequence_parallel_gpu.py -q -m gpu > $O/tests_d.log 2>&1; tail -4 $O/tests_d.log
python bench.py --steps 10 --warmup 2 --no-cpu-baseline > $O/bench_fold_on.json.log 2> $O/bench_fold_on.err; tail -c 300 $O/bench_fold_on.json.log
GF_FOLD_PAD_KEYS=0 python bench.py --steps 10 --warmup 2 --no-cpu-baseline > $O/bench_fold_off.json.log 2> $O/bench_fold_off.err; tail -c 300 $O/bench_fold_off.json.log
