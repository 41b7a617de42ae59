#!/bin/bash
# Round 4, GPU batch C: the whole GPU suite on the final library (ORMAX attention, Q'-scaled backward, Canny), then batch B.
O=gpurun_out/r04
mkdir -p $O
python -m pytest tests -q -m gpu > $O/gpu_suite.log 2>&1; tail -5 $O/gpu_suite.log
python tools/attn_ab.py --rounds 4 > $O/attn_ab_final.log 2>&1; cat $O/attn_ab_final.log
python tools/attn_clock.py > $O/attn_clock.log 2>&1; python tools/attn_clock.py --zeros >> $O/attn_clock.log 2>&1; cat $O/attn_clock.log
bash tools/gpu_batch_r04_b.sh
