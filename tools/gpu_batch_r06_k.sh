#!/bin/bash
# round 6, call K: the GPU suite and the driver's bench command on the tree with the pipeline as an nn.Module, the two-rank training loop,
# the training launcher; smoke
O=gpurun_out/r06
mkdir -p $O
( time timeout 1500 python -m pytest tests -m gpu -q -x --durations=15 ) > $O/gpu_suite_k.log 2>&1
echo "suite rc=$?" >> $O/gpu_suite_k.log
tail -24 $O/gpu_suite_k.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke_k.log 2>&1; tail -2 $O/smoke_k.log | cut -c1-300
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/bench_driver_cmd_k.json.log 2> $O/bench_driver_cmd_k.stderr.log
echo "driver cmd rc=$?"; grep "real\|yardstick" $O/bench_driver_cmd_k.stderr.log | cut -c1-250
python3 - <<'PY'
import json
j=json.loads([l for l in open('gpurun_out/r06/bench_driver_cmd_k.json.log') if l.startswith('{')][0])
print({k:j[k] for k in ('value','ms_per_step','denoise_step_ms_high_noise','denoise_step_ms_low_noise','vae_decode_s')}, j['roofline']['frac'], j['roofline']['avg_launch_ms'])
print('latents', j['self_check']['latents']['sha256'][:12], 'frames', j['self_check']['frames_uint8']['sha256'][:12], 'yardstick x', j['gpu_eager_yardstick'].get('speedup_denoise_loop'))
PY
