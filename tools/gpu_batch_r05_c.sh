#!/bin/bash
# round 5, call C: the VAE on the padded-layout kernels — tests, timings, per-kernel traces, counters of the three dominant conv launches
O=gpurun_out/r05
mkdir -p $O
export TMPDIR=/tmp
( timeout 1200 python -m pytest tests/test_vae.py tests/test_preloop.py tests/test_e2e_gpu.py tests/test_sequence_parallel_gpu.py -m gpu -x -q ) > $O/vae_tests_c.log 2>&1
echo "tests rc=$?" >> $O/vae_tests_c.log; tail -4 $O/vae_tests_c.log
( timeout 600 python tools/conv_a4_bench.py ) 2>&1 | grep -v "^MIOpen\|amdgpu.ids" > $O/conv_a4_bench_c.log; grep "up2x" $O/conv_a4_bench_c.log
( timeout 300 python tools/vae_profile.py decode 3; timeout 300 python tools/vae_profile.py encode 3 ) 2>&1 | grep -v "^MIOpen\|amdgpu.ids" > $O/vae_times_c.log; cat $O/vae_times_c.log
for what in decode encode; do
  rocprofv3 --kernel-trace --stats -d $O/trace_vae_$what -o v -- python3 tools/vae_profile.py $what 1 > $O/trace_vae_$what.log 2>&1
  python3 tools/rocpd_table.py $O/trace_vae_$what/v_results.db 30 > $O/vae_${what}_by_kernel_and_grid.md
  rm -rf $O/trace_vae_$what
  head -24 $O/vae_${what}_by_kernel_and_grid.md | cut -c1-170
done
ONLY=vae bash tools/profile_r05.sh > $O/pmc_vae.log 2>&1
cat $O/pmc/vae_l2_FETCH_SIZE.md $O/pmc/vae_l2_WRITE_SIZE.md | cut -c1-200
