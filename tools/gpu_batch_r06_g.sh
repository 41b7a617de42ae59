#!/bin/bash
# round 6, call G: the GPU suite on the widened tree (training-mode datasets, pixel round trip, masked control videos), durations
O=gpurun_out/r06
mkdir -p $O
( time timeout 1500 python -m pytest tests -m gpu -q -x --durations=15 ) > $O/gpu_suite_g.log 2>&1
echo "suite rc=$?" >> $O/gpu_suite_g.log
tail -28 $O/gpu_suite_g.log
