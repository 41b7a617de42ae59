#!/bin/bash
# round 6, call N: the VAE attention's two-GEMM scores (gf_rowmax_neg_bf16) — its own tests and every test that goes through the VAE
O=gpurun_out/r06
mkdir -p $O
( time timeout 900 python -m pytest tests/test_vae.py tests/test_preloop.py tests/test_pipeline_call_gpu.py -m gpu -q -x -s --durations=8 ) > $O/vae_attn_n.log 2>&1
echo "rc=$?" >> $O/vae_attn_n.log
grep "VAE attention\|passed\|failed\|Error\|rc=" $O/vae_attn_n.log | cut -c1-250 | tail -30
python3 - > $O/vae_attn_time_n.log 2>&1 <<'PY'
import torch
from goal_force_amd import ops, vae
C, hw, g = 384, 1560, 4
qkv = torch.randn((g, hw, 3 * C), device="cuda").to(torch.bfloat16)
out = torch.empty((g, hw, C), dtype=torch.bfloat16, device="cuda")
for flag in (True, False, True, False):
    with ops.options(vae_attn_offset=flag):
        for _ in range(3): vae.frame_attention(qkv, C, out)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): vae.frame_attention(qkv, C, out)
        e1.record(); torch.cuda.synchronize()
        print(f"frame_attention g={g} hw={hw} C={C} two_gemm={flag}: {e0.elapsed_time(e1) / 20:.3f} ms")
PY
cat $O/vae_attn_time_n.log | cut -c1-200
