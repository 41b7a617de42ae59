#!/bin/bash
# round 6, call D: the driver's command on a fresh box (with the eager yardstick leg), the rocprofv3 kernel trace of the bench command
# (per-kernel table the roofline figure must agree with), the attention kernel's PMC passes on this round's library, smoke(), the GPU suite
O=gpurun_out/r06
mkdir -p $O/pmc
export TMPDIR=/tmp
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/bench_driver_cmd_d.json.log 2> $O/bench_driver_cmd_d.stderr.log
echo "driver cmd rc=$?"; grep "real\|yardstick" $O/bench_driver_cmd_d.stderr.log | cut -c1-250
rocprofv3 --kernel-trace --stats -d $O/trace_d -o bench -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --config5-steps 0 --peaky-steps 0 --no-preloop --no-yardstick > $O/bench_steps5.json.log 2> $O/trace_d.err
python3 tools/rocpd_table.py $O/trace_d/bench_results.db 40 > $O/bench_steps5_by_kernel_and_grid.md 2>> $O/trace_d.err
head -14 $O/bench_steps5_by_kernel_and_grid.md | cut -c1-220
rm -rf $O/trace_d
pass() {   # pass <tag> <counters> <kernel filter> <program> <args...>
  local tag=$1 ctr=$2 filt=$3 prog=$4; shift 4
  rocprofv3 --pmc $ctr --kernel-trace -d $O/pmc/raw_$tag -o a -- python3 $prog "$@" > $O/pmc/$tag.log 2>&1
  python3 tools/rocpd_pmc.py $O/pmc/raw_$tag/a_results.db "$filt" > $O/pmc/$tag.md 2>> $O/pmc/err.log
  rm -rf $O/pmc/raw_$tag
}
SQ1="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE"
pass attn_FETCH_SIZE FETCH_SIZE "flash_attn|transpose_v32" tools/microbench.py attn --iters 2
pass attn_WRITE_SIZE WRITE_SIZE "flash_attn|transpose_v32" tools/microbench.py attn --iters 2
pass attn_SQ "$SQ1" "flash_attn" tools/microbench.py attn --iters 2
head -8 $O/pmc/attn_FETCH_SIZE.md $O/pmc/attn_WRITE_SIZE.md | cut -c1-200
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -2 $O/smoke.log | cut -c1-300
( time timeout 1500 python -m pytest tests -m gpu -q -x --durations=25 ) > $O/gpu_suite_d.log 2>&1
echo "suite rc=$?" >> $O/gpu_suite_d.log
tail -34 $O/gpu_suite_d.log
