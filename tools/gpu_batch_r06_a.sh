#!/bin/bash
# round 6, call A: the GPU suite on the tree without env selectors, with the full-size N = 2 CFG-pair test; full durations list
O=gpurun_out/r06
mkdir -p $O
( time timeout 1500 python -m pytest tests -m gpu -q -x --durations=45 ) > $O/gpu_suite_a.log 2>&1
echo "suite rc=$?" >> $O/gpu_suite_a.log
tail -60 $O/gpu_suite_a.log
