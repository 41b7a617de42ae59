#!/bin/bash
# round 5, call D: after the per-convolution dispatch and the plain-store epilogue — conv A/B, VAE times, counters, then the WHOLE suite and the driver's bench
O=gpurun_out/r05
mkdir -p $O
export TMPDIR=/tmp
( timeout 600 python tools/conv_a4_bench.py ) 2>&1 | grep -v "^MIOpen\|amdgpu.ids" > $O/conv_a4_bench_d.log; cat $O/conv_a4_bench_d.log
( timeout 300 python tools/vae_profile.py decode 3; timeout 300 python tools/vae_profile.py encode 3 ) 2>&1 | grep -v "^MIOpen\|amdgpu.ids" > $O/vae_times_d.log; cat $O/vae_times_d.log
ONLY=vae bash tools/profile_r05.sh > $O/pmc_vae.log 2>&1
cat $O/pmc/vae_l2_FETCH_SIZE.md $O/pmc/vae_l2_WRITE_SIZE.md | grep conv | cut -c1-200
( time timeout 1500 python -m pytest tests -m gpu -x -q --durations=8 ) > $O/gpu_suite_d.log 2>&1
echo "suite rc=$?" >> $O/gpu_suite_d.log; tail -14 $O/gpu_suite_d.log
( time timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/bench_driver_cmd_d.json.log 2> $O/bench_driver_cmd_d.stderr.log
echo "bench rc=$?" >> $O/bench_driver_cmd_d.stderr.log; tail -4 $O/bench_driver_cmd_d.stderr.log
