#!/bin/bash
# round 6, call F: the masked control videos on the GPU; the attention kernel on adversarial data (logits rising along the key index so that the
# lazy rescale fires in every tile) next to random data, interleaved, with an fp64 check
O=gpurun_out/r06
mkdir -p $O
( timeout 600 python -m pytest tests/test_force_map.py -m gpu -q -x ) > $O/force_map_f.log 2>&1; tail -3 $O/force_map_f.log
( timeout 600 python tools/microbench.py attn --iters 6 --ramp 8 ) > $O/attn_ramp8.log 2>&1; tail -3 $O/attn_ramp8.log | cut -c1-400
( timeout 600 python tools/microbench.py attn --iters 6 --ramp 2 ) > $O/attn_ramp2.log 2>&1; tail -2 $O/attn_ramp2.log | cut -c1-400
( timeout 600 python tools/microbench.py attn --iters 6 --ramp 40 ) > $O/attn_ramp40.log 2>&1; tail -2 $O/attn_ramp40.log | cut -c1-400
