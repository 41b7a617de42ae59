"""CPU ORACLE — TEST INFRASTRUCTURE ONLY.  Restatement of AutoWrappedLinear.fp8_linear
(diffsynth/vram_management/layers.py:115-151) for computation_dtype = float8_e4m3fn (OCP; gfx950 native):

    x_max   = max|x| per row;  scale_a = clamp(x_max / 448, min=1).float();  scale_b = ones
    x8      = (x / (scale_a + 1e-8)).to(float8_e4m3fn);   w8 = w.to(float8_e4m3fn);  bias -> bf16
    result  = torch._scaled_mm(x8, w8.T, scale_a, scale_b.T, bias, out_dtype=x.dtype)

torch._scaled_mm needs a GPU, so the product of the quantised operands is evaluated here in fp32 (every fp8 x fp8
product is exact in fp32; only the accumulation order is unspecified) — "parity unpinned" against a reference
run, pinned instead by the hand-computed known-answer vectors in tests/test_fp8.py."""
import torch

FP8 = torch.float8_e4m3fn


def quantize_activation(x):
    x2 = x.reshape(-1, x.shape[-1])
    x_max = torch.max(torch.abs(x2), dim=-1, keepdim=True).values
    scale_a = torch.clamp(x_max / 448.0, min=1.0).float()
    x8 = (x2 / (scale_a + 1e-8)).to(FP8)
    return x8, scale_a


def fp8_linear(x, weight, bias):
    x8, scale_a = quantize_activation(x)
    w8 = weight.to(FP8)
    acc = x8.float() @ w8.float().t()
    out = (acc * scale_a * 1.0 + bias.to(torch.bfloat16).float()).to(x.dtype)
    return out.reshape(x.shape[:-1] + (weight.shape[0],))
