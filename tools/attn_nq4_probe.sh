#!/bin/bash
# Builds (CPU container) and runs (GPU box) the timing-only 64-rows-per-wave attention loop and its what-if variants.
#   bash tools/attn_nq4_probe.sh build | run
set -e
cd "$(dirname "$0")/.."
if [ "$1" = "build" ]; then
  mkdir -p build/nq4
  for v in full:0:0:0:dma novalu:1:0:0:dma nodma:0:1:0:dma noreads:0:0:1:dma mfmaonly:1:1:1:dma reg:0:0:0:reg reg_novalu:1:0:0:reg; do
    IFS=: read name a b c st <<< "$v"
    NQ4_STAGE=$st NQ4_NO_VALU=$a NQ4_NO_DMA=$b NQ4_NO_READS=$c python3 tools/gen_attn_nq4.py > build/nq4/attn_nq4_loop.inc 2>/dev/null
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -Wno-unused-value -Ibuild/nq4 -o build/nq4/attn_nq4_$name tools/probes/attn_nq4_whatif.hip
  done
  # placement variants of the register staging: loads packed / spread, writes early / late
  for v in reg_ld8:2:8:100:4 reg_wr64:2:4:64:8 reg_wr40:2:4:40:12 reg_ld16:2:16:124:1; do
    IFS=: read name l0 ls w0 ws <<< "$v"
    NQ4_STAGE=reg NQ4_LD0=$l0 NQ4_LDS=$ls NQ4_WR0=$w0 NQ4_WRS=$ws python3 tools/gen_attn_nq4.py > build/nq4/attn_nq4_loop.inc 2>/dev/null
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -Wno-unused-value -Ibuild/nq4 -o build/nq4/attn_nq4_$name tools/probes/attn_nq4_whatif.hip
  done
  ls -la build/nq4
else
  for name in full novalu nodma noreads mfmaonly reg reg_novalu reg_ld8 reg_wr64 reg_wr40 reg_ld16; do
    echo "== $name"; timeout 120 ./build/nq4/attn_nq4_$name random ${2:-512}; timeout 120 ./build/nq4/attn_nq4_$name zeros ${2:-512}
  done
fi
