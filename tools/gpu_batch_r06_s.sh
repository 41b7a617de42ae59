#!/bin/bash
# round 6, call S: the new backward-precision tests; the training step at production size with every self-attention's logits x 3, pre-scaled vs plain q
O=gpurun_out/r06
mkdir -p $O
( time timeout 600 python -m pytest tests/test_training_gpu.py -m gpu -q -x -s -k "peaky" ) > $O/train_tests_s.log 2>&1
echo "rc=$?" >> $O/train_tests_s.log; grep "attention backward\|block backward\|passed\|failed\|Error\|rc=" $O/train_tests_s.log | cut -c1-900
( time timeout 1300 python tests/fullsize_train_parity.py --peaky 3 --out $O/fullsize_train_parity_peaky3.json ) > $O/fullsize_train_parity_peaky3.log 2>&1
echo "rc=$?" >> $O/fullsize_train_parity_peaky3.log; tail -14 $O/fullsize_train_parity_peaky3.log | cut -c1-900
