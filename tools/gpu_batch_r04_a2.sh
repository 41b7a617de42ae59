#!/bin/bash
# Round 4, GPU batch A2: where the Canny kernels and the numpy restatement part; the tests batch A did not reach; second attention A/B.
O=gpurun_out/r04
mkdir -p $O
python tools/canny_debug.py > $O/canny_debug.log 2>&1; cat $O/canny_debug.log
python -m pytest tests/test_canny.py -q -m gpu > $O/tests_canny.log 2>&1; tail -4 $O/tests_canny.log
python -m pytest tests/test_kernels_gpu.py tests/test_training_gpu.py tests/test_vae.py -q -m gpu -s > $O/tests_a2.log 2>&1; tail -3 $O/tests_a2.log; grep "SiLU C=" $O/tests_a2.log
GOALFORCE_HIP_LIB=$PWD/build/ab/libgoalforce_ormax.so python -m pytest tests/test_kernels_gpu.py tests/test_dit_gpu.py -q -m gpu -k "flash_attn or attention or block or model_fn or loop" > $O/tests_ormax2.log 2>&1; tail -2 $O/tests_ormax2.log
python tools/attn_ab.py --rounds 5 > $O/attn_ab_r04.log 2>&1; cat $O/attn_ab_r04.log
python tools/attnbwd_ab.py > $O/attnbwd_ab_qscale.log 2>&1; cat $O/attnbwd_ab_qscale.log
python tools/energy_probe.py 5 > $O/attn_energy_probe.log 2>&1; head -48 $O/attn_energy_probe.log
