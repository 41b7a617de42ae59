#!/bin/bash
# Round 4, GPU batch A (run on the GPU box from the repo root): new tests, the ORMAX attention A/B, the in-kernel clock, backward timing.
O=gpurun_out/r04
mkdir -p $O
python -m pytest tests/test_canny.py tests/test_kernels_gpu.py tests/test_training_gpu.py tests/test_vae.py -q -m gpu -x -s > $O/tests_a.log 2>&1; tail -3 $O/tests_a.log
python -m pytest tests/test_bench_gpu.py -q -m gpu -x > $O/tests_bench.log 2>&1; tail -3 $O/tests_bench.log
GOALFORCE_HIP_LIB=$PWD/build/ab/libgoalforce_ormax.so python -m pytest tests/test_kernels_gpu.py tests/test_dit_gpu.py -q -m gpu -x -k "flash_attn or attention or block or model_fn or loop" > $O/tests_ormax.log 2>&1; tail -3 $O/tests_ormax.log
python tools/attn_ab.py --rounds 5 > $O/attn_ab_ormax.log 2>&1; cat $O/attn_ab_ormax.log
python tools/attn_ab.py --rounds 3 --zeros > $O/attn_ab_ormax_zeros.log 2>&1; cat $O/attn_ab_ormax_zeros.log
python tools/attn_clock.py > $O/attn_clock.log 2>&1; python tools/attn_clock.py --zeros >> $O/attn_clock.log 2>&1; cat $O/attn_clock.log
python tools/microbench.py attnbwd --iters 8 > $O/microbench_attnbwd_qscale.log 2>&1; cat $O/microbench_attnbwd_qscale.log
python tools/energy_probe.py 6 > $O/attn_energy_probe.log 2>&1; head -40 $O/attn_energy_probe.log
