#!/bin/bash
# round 6, call P: umT5-XXL at full size against the reference's arithmetic on torch-ROCm; the new suite tests (VAE all tiles, umT5 full size)
O=gpurun_out/r06
mkdir -p $O
( time timeout 600 python tests/fullsize_t5_parity.py --out $O/fullsize_t5_parity.json ) > $O/fullsize_t5_parity.log 2>&1
echo "rc=$?" >> $O/fullsize_t5_parity.log; tail -8 $O/fullsize_t5_parity.log | cut -c1-700
( time timeout 600 python -m pytest tests/test_vae.py tests/test_text_encoder.py -m gpu -q -x -s --durations=5 -k "full_size" ) > $O/new_tests_p.log 2>&1
echo "rc=$?" >> $O/new_tests_p.log; tail -25 $O/new_tests_p.log | cut -c1-700
