#!/bin/bash
# round 6, call Q: the ControlNet training step at production size (40 + 10 blocks, 32760 tokens) against the reference's arithmetic under
# autograd on torch-ROCm (bf16 and fp32, block-wise recomputation); a quick look at 4 + 2 blocks first
O=gpurun_out/r06
mkdir -p $O
( time timeout 400 python tests/fullsize_train_parity.py --layers 2 --cn-layers 1 --frames 2 --out $O/train_parity_quick.json ) > $O/train_parity_quick.log 2>&1
rc=$?; echo "rc=$rc" >> $O/train_parity_quick.log; tail -12 $O/train_parity_quick.log | cut -c1-600
[ $rc -eq 0 ] || exit 0
( time timeout 1300 python tests/fullsize_train_parity.py --out $O/fullsize_train_parity.json ) > $O/fullsize_train_parity.log 2>&1
echo "rc=$?" >> $O/fullsize_train_parity.log; tail -14 $O/fullsize_train_parity.log | cut -c1-900
