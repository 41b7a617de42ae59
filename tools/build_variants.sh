#!/bin/bash
# A/B and diagnostic builds of the attention kernels (CPU container; the .so files under build/ab/ travel to the GPU box):
#   libgf_base.so / libgf_max3watch.so / libgf_naivemap.so   tools/attn_ab.py   kernel 3 as shipped / with the ORMAX watch (round 4, not shipped) / plain head-major block order
#   gfclock.so                       tools/attn_clock.py       kernel 3 with in-kernel clock stamps
#   libgoalforce_ormax.so            GOALFORCE_HIP_LIB=...     the whole library with the ORMAX attention watch (runs the test suite on it)
set -e
cd "$(dirname "$0")/.."
make -C goal_force_amd/csrc -j4 > /dev/null
bash tools/experimental_tree.sh        # the variants below are selected by macros that only the experimental tree has (tools/patches/)
X=build/experimental
python3 tools/attn_ab.py --build base: ormax:-DGF_K3_ORMAX=1 naivemap:-DGF_K3_MAP=1
python3 tools/attnbwd_ab.py --build
python3 tools/attn_clock.py --build
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -I$X/include -I$X/csrc -Wall -Wno-unused-function -fvisibility=hidden -DGF_BUILD"
/opt/rocm/bin/hipcc $F -mllvm -amdgpu-mfma-vgpr-form -DGF_K3_ORMAX=1 -c $X/csrc/gf_attention.hip -o build/ab/gf_attention_ormax.o
/opt/rocm/bin/hipcc $F -c $X/csrc/gf_abi.hip -o build/ab/gf_abi_x.o
OBJS=$(ls build/csrc/gf_*.o | grep -v -e "gf_attention.o" -e "gf_gemm-" -e "gf_abi.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/ab/libgoalforce_ormax.so $OBJS build/ab/gf_attention_ormax.o build/ab/gf_abi_x.o
ls -la build/ab/
