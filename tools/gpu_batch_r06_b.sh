#!/bin/bash
# round 6, call B: the new tests (g13 whole-call golden, canny launcher, bench line with the eager yardstick), where the 8-rank launch
# spends its wall time (phase log), and the driver's command with the yardstick leg
O=gpurun_out/r06
mkdir -p $O
( time timeout 900 python -m pytest tests/test_pipeline_call_gpu.py tests/test_e2e_gpu.py tests/test_vae.py tests/test_kernels_gpu.py::test_cross_attention_folds_the_padded_context_rows tests/test_training_gpu.py::test_dit_block_backward_is_the_same_whatever_the_forward_kept "tests/test_bench_gpu.py::test_bench_line_is_schedule_weighted_and_carries_preloop_vae_roofline_and_data_sensitivity" -m gpu -q -x -s --durations=12 ) > $O/new_tests_b.log 2>&1
echo "rc=$?" >> $O/new_tests_b.log
tail -25 $O/new_tests_b.log
( time GF_DIST_BACKEND=gloo timeout 600 python bench.py --gpus 8 --layers 1 --steps 1 --warmup 0 --no-cpu-baseline ) > $O/bench_n8_phases.json.log 2> $O/bench_n8_phases.stderr.log
echo "rc=$?"; grep "bench.py \[\|real" $O/bench_n8_phases.stderr.log | cut -c1-200
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/bench_driver_cmd_b.json.log 2> $O/bench_driver_cmd_b.stderr.log
echo "rc=$?"; grep "bench.py \[\|real\|yardstick" $O/bench_driver_cmd_b.stderr.log | cut -c1-250
python3 - <<'PY'
import json
j=json.loads([l for l in open('gpurun_out/r06/bench_driver_cmd_b.json.log') if l.startswith('{')][0])
print({k:j[k] for k in ('value','ms_per_step','denoise_step_ms_high_noise','denoise_step_ms_low_noise','vae_decode_s')})
print('roofline', j['roofline']['frac'], j['roofline']['avg_launch_ms'], j['roofline']['traffic'])
print('yardstick', j.get('gpu_eager_yardstick'))
print('latents', j['self_check']['latents']['sha256'][:12], 'frames', j['self_check']['frames_uint8']['sha256'][:12])
PY
