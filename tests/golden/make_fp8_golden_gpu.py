"""Generates tests/golden/g11_fp8_scaled_mm.npz ON A GPU BOX (torch._scaled_mm has no CPU kernel):

    gpurun -- 'python tests/golden/make_fp8_golden_gpu.py gpurun_out/g11_fp8_scaled_mm.npz'

The reference's fp8 contract IS a torch call sequence (AutoWrappedLinear.fp8_linear, diffsynth/vram_management/
layers.py:115-151, computation_dtype = float8_e4m3fn); /root/reference does not exist on the GPU box, so the sequence
is restated here line by line from that citation (row max -> clamp(x_max/448, min=1).float() -> x/(scale_a+1e-8) ->
.to(e4m3fn); weight.to(e4m3fn); torch._scaled_mm(x8, w8.T, scale_a, ones(N,1).T, bias.bf16, out_dtype=bf16)) and run
through torch's own kernels.  Stored: the quantised activations (uint8 bit patterns), scale_a, the quantised weight and
the _scaled_mm output (bf16 bits) for the seeded cases of gen_inputs.fp8_case — the pin for oracle/fp8_oracle.py
(tests/test_fp8.py::test_oracle_vs_scaled_mm_fixture) and for the HIP kernels (test_hip_fp8_vs_torch_scaled_mm)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import gen_inputs as gi  # noqa: E402


def scaled_mm_linear(x, weight, bias):
    """VRAM:115-151 on the device, fp8 dtype = torch.float8_e4m3fn."""
    shape = x.shape
    x = x.reshape(-1, shape[-1])
    x_max = torch.max(torch.abs(x), dim=-1, keepdim=True).values
    scale_a = torch.clamp(x_max / 448.0, min=1.0).float().to(device=x.device)
    scale_b = torch.ones((weight.shape[0], 1)).to(device=x.device)
    x8 = (x / (scale_a + 1e-8)).to(torch.float8_e4m3fn)
    w8 = weight.to(torch.float8_e4m3fn)
    out = torch._scaled_mm(x8, w8.T, scale_a=scale_a, scale_b=scale_b.T, bias=bias.to(torch.bfloat16), out_dtype=x.dtype)
    return out.reshape(shape[:-1] + out.shape[-1:]), x8, scale_a, w8


if __name__ == "__main__":
    path = sys.argv[1]
    out = {"torch": np.array(torch.__version__), "device": np.array(torch.cuda.get_device_name(0))}
    for i, (M, N, K) in enumerate(gi.FP8_CASES):
        x, w, b = gi.fp8_case(M, N, K)
        y, x8, s, w8 = scaled_mm_linear(x.cuda(), w.cuda(), b.cuda())
        out[f"y{i}"] = gi.to_u16(y.cpu())
        out[f"x8_{i}"] = x8.cpu().view(torch.uint8).numpy()
        out[f"s{i}"] = s.cpu().numpy()
        out[f"w8_{i}"] = w8.cpu().view(torch.uint8).numpy() if N * K <= 1 << 20 else np.zeros(0, np.uint8)
        out[f"ck{i}"] = np.array(gi.checksum([x, w, b]))
        print(f"case {i} {M}x{N}x{K}: scale_a max {float(s.max()):.3f}, |y| mean {float(y.float().abs().mean()):.4f}")
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB")
