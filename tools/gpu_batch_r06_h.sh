#!/bin/bash
# round 6, call H: the training launcher end to end; the two-rank head-parallel test under the suite's thread budget
O=gpurun_out/r06
mkdir -p $O
( time timeout 1200 python -m pytest tests/test_e2e_gpu.py tests/test_sequence_parallel_gpu.py -m gpu -q -x --durations=8 ) > $O/new_tests_h.log 2>&1
echo "rc=$?" >> $O/new_tests_h.log
grep -v "^$" $O/new_tests_h.log | tail -40
