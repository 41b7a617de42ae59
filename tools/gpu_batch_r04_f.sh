#!/bin/bash
# Round 4, batch F (GPU box, repo root): the attention watch is back to round 3's v_max3 form (ORMAX measured slower on peaky logits), so:
# attention tests first, the attention counter passes again -> profiles/pmc_static.json (GEMM tables unchanged, copied in), A/B, the whole
# suite, the driver's bench command, the 4-step full-size parity on this build.
O=gpurun_out/r04
mkdir -p $O/pmc
export TMPDIR=/tmp
cp profiles/r04/pmc/gemm_*_SIZE.md $O/pmc/
ONLY=attn bash tools/profile_r04.sh > $O/profile_r04_f.log 2>&1; tail -2 $O/profile_r04_f.log
python tools/pmc_static.py $O/pmc profiles/r04/pmc > $O/pmc_static.log 2>&1; cat $O/pmc_static.log; cp profiles/pmc_static.json $O/pmc_static.json
python tools/attn_ab.py --rounds 4 > $O/attn_ab_final2.log 2>&1; cat $O/attn_ab_final2.log
python -m pytest tests -q -m gpu > $O/gpu_suite_final2.log 2>&1; tail -4 $O/gpu_suite_final2.log
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd_final.json.log 2> $O/bench_driver_cmd_final.err; tail -c 600 $O/bench_driver_cmd_final.json.log; tail -3 $O/bench_driver_cmd_final.err
python bench.py --steps 4 --warmup 1 --no-cpu-baseline --config5-steps 0 --peaky 8 > $O/bench_final_peaky8.json.log 2>&1
timeout 1200 python tests/fullsize_parity.py --steps 4 --fp8 --out $O/fullsize_parity_4step_final2.json > $O/fullsize_parity_4step_final2.log 2>&1; tail -8 $O/fullsize_parity_4step_final2.log | cut -c1-400
