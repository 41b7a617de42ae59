#!/bin/bash
# round 6, call E: the widened rows' GPU tests (training loop vs g17, forward_preprocess vs g16, Plants / Dominos vs g15)
O=gpurun_out/r06
mkdir -p $O
( time timeout 1200 python -m pytest tests/test_training_gpu.py tests/test_preloop.py tests/test_force_map.py -m gpu -q -x -s --durations=8 ) > $O/new_tests_e.log 2>&1
echo "rc=$?" >> $O/new_tests_e.log
grep -v "^$" $O/new_tests_e.log | tail -40
