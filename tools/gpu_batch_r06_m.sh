#!/bin/bash
# round 6, call M: parity at production size on PEAKY attention (every self-attention's logits x 3 and x 8: near-one-hot softmax rows, the
# regime of trained weights) — one full forward (40 + 10 blocks, S = 32760) against fp32 math next to the reference's bf16 arithmetic, then a
# 4-step CFG loop at x 8 with the decoded frames
O=gpurun_out/r06
mkdir -p $O
for f in 3 8; do
  ( timeout 900 python tests/fullsize_parity.py --steps 0 --peaky $f --out $O/fullsize_forward_peaky$f.json ) > $O/fullsize_forward_peaky$f.log 2>&1
  echo "peaky $f rc=$?"; grep "vs fp32\|vs ref" $O/fullsize_forward_peaky$f.log | cut -c1-330
done
( time timeout 1500 python tests/fullsize_parity.py --steps 4 --peaky 8 --out $O/fullsize_parity_4step_peaky8.json ) > $O/fullsize_parity_4step_peaky8.log 2>&1
echo "loop rc=$?"; grep -v "oracle\[" $O/fullsize_parity_4step_peaky8.log | grep "vs fp32\|PSNR\|real" | cut -c1-400
