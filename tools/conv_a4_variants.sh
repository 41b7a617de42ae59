#!/bin/bash
# Schedule variants of the padded-layout convolution's K loop (tools/gen_conv_a4.py: CONV_A4_DS / CONV_A4_WAIT / CONV_A4_RDSTEP), each as a
# whole library under build/variants/conv_<name>/libgoalforce_hip.so (CPU container; the .so files travel to the GPU box).  On the box:
#   for d in build/variants/conv_*; do GOALFORCE_HIP_LIB=$d/libgoalforce_hip.so python tools/conv_a4_bench.py; done
# Nothing built here is loaded by goal_force_amd unless GOALFORCE_HIP_LIB points at it.
set -e
cd "$(dirname "$0")/.."
make -C goal_force_amd/csrc -j8 > /dev/null
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude -Wall -Wno-unused-function -fvisibility=hidden -DGF_BUILD"
OBJS=$(ls build/csrc/gf_*.o | grep -v -e "gf_conv_a4.o" -e "gf_gemm-" -e "gf_conv_a4-")
for v in "${@:-ds4_w64_r2:4:64:2 ds4_w64_r1:4:64:1 ds4_w70_r1:4:70:1 ds5_w70_r1:5:70:1 ds5_w76_r1:5:76:1 ds6_w78_r1:6:78:1 ds4_w58_r2:4:58:2}"; do
  for spec in $v; do
    IFS=: read name ds w r <<< "$spec"
    d=build/variants/conv_$name
    mkdir -p $d
    cp goal_force_amd/csrc/gf_conv_a4.hip goal_force_amd/csrc/gf_common.h $d/
    CONV_A4_DS=$ds CONV_A4_WAIT=$w CONV_A4_RDSTEP=$r CONV_A4_OUT=$d/gf_conv_a4_loop.inc python3 tools/gen_conv_a4.py > $d/gen.log
    /opt/rocm/bin/hipcc $F -I$d -c $d/gf_conv_a4.hip -o $d/gf_conv_a4.o 2> $d/cc.log
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $d/libgoalforce_hip.so $OBJS $d/gf_conv_a4.o
    echo "$d: $(tail -1 $d/gen.log | cut -c1-160)"
  done
done
