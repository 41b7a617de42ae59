#!/bin/bash
# round 5, call J (the round's last tree): the driver's bench command, then the complete 50-step CFG loop at production size against the
# fp32 oracle, bf16 and fp8 (100 fp32 forwards: ~50 GPU-minutes) — the loop-level parity record of this round's build
O=gpurun_out/r05
mkdir -p $O
( time timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/bench_driver_cmd_final.json.log 2> $O/bench_driver_cmd_final.stderr.log
echo "bench rc=$?" >> $O/bench_driver_cmd_final.stderr.log; tail -4 $O/bench_driver_cmd_final.stderr.log
( time timeout 4500 python tests/fullsize_parity.py --steps 50 --fp8 --out $O/fullsize_parity_50step.json ) > $O/fullsize_parity_50step.log 2>&1
grep -v "^MIOpen\|amdgpu.ids" $O/fullsize_parity_50step.log | grep "PSNR\|real" | cut -c1-400
