#!/bin/bash
# Builds (CPU container) and runs (GPU box) the timing-only 64-rows-per-wave attention loop and its what-if variants.
#   bash tools/attn_nq4_probe.sh build | run
set -e
cd "$(dirname "$0")/.."
if [ "$1" = "build" ]; then
  mkdir -p build/nq4
  for v in full:0:0:0 novalu:1:0:0 nodma:0:1:0 noreads:0:0:1 mfmaonly:1:1:1; do
    IFS=: read name a b c <<< "$v"
    NQ4_NO_VALU=$a NQ4_NO_DMA=$b NQ4_NO_READS=$c python3 tools/gen_attn_nq4.py > build/nq4/attn_nq4_loop.inc 2>/dev/null
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -Wno-unused-value -Ibuild/nq4 -o build/nq4/attn_nq4_$name tools/probes/attn_nq4_whatif.hip
  done
  ls -la build/nq4
else
  for name in full novalu nodma noreads mfmaonly; do
    echo "== $name"; timeout 120 ./build/nq4/attn_nq4_$name random ${2:-512}; timeout 120 ./build/nq4/attn_nq4_$name zeros ${2:-512}
  done
fi
