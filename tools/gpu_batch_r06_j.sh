#!/bin/bash
# round 6, call J (run twice: before and after the pre-scaled Q; the second run writes *_final): the complete 50-step CFG loop at production size on this round's tree, bf16 and fp8, against the reference arithmetic on
# torch-ROCm step by step (tests/fullsize_parity.py; 51 GPU-minutes in round 5) — the device code of the step is round 5's, so every figure
# should repeat to the last digit; what is new underneath is the host side (no env selectors, option plumbing, VAE pool scope)
O=gpurun_out/r06
mkdir -p $O
( time timeout 4200 python tests/fullsize_parity.py --steps 50 --fp8 --out $O/fullsize_parity_50step_final.json ) > $O/fullsize_parity_50step_final.log 2>&1
echo "rc=$?" >> $O/fullsize_parity_50step_final.log
grep -v "oracle\[" $O/fullsize_parity_50step_final.log | tail -12 | cut -c1-600
