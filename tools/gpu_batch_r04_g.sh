#!/bin/bash
# Round 4, batch G (GPU box, repo root): the complete 50-step CFG loop at full size, bf16 + fp8, on the shipped build.
O=gpurun_out/r04
mkdir -p $O
timeout 3500 python tests/fullsize_parity.py --steps 50 --fp8 --out $O/fullsize_parity_50step_shipped.json > $O/fullsize_parity_50step_shipped.log 2>&1
tail -6 $O/fullsize_parity_50step_shipped.log | cut -c1-600
