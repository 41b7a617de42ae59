#!/bin/bash
# Package power and shader clock (rocm-smi, once a second) while a kernel loops: is the chip at its power limit under this kernel?
#   bash tools/power_sample.sh attn|gemm   -> gpurun_out/power_<what>.log
what=${1:-attn}
out=gpurun_out/power_$what.log
mkdir -p gpurun_out
rocm-smi --showmaxpower 2>&1 | grep -i "Max Graphics" > $out
python3 tools/microbench.py $what --iters 1200 > gpurun_out/power_${what}_bench.log 2>&1 &
pid=$!
for i in $(seq 1 60); do
  sleep 1
  kill -0 $pid 2>/dev/null || break
  echo "t=${i}s $(rocm-smi --showpower --showclocks 2>&1 | grep -i 'Package Power\|sclk' | sed 's/GPU\[0\]\t*: //' | tr '\n' ' ')" >> $out
done
wait $pid
grep -v amdgpu.ids gpurun_out/power_${what}_bench.log | tail -3 >> $out
cat $out
