#!/usr/bin/env python3
"""One VAE convolution shape of a production tile, a few launches — the program behind `rocprofv3 --pmc ... -- python3 tools/conv_pmc.py <shape>`
(tools/profile_r05.sh: one shape per pass, so that a counter table row is one shape).
shapes: l2 = 3x3x3 192->192 over 81 x 120 x 208 (+residual), l1 = 384->384 over 41 x 60 x 104, c96 = 96->96 over 81 x 240 x 416."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from goal_force_amd import ops  # noqa: E402

BF = torch.bfloat16
SHAPES = {"l2": (81, 120, 208, 192, 192), "l1": (41, 60, 104, 384, 384), "c96": (81, 240, 416, 96, 96)}


def main():
    T, H, W, C, N = SHAPES[sys.argv[1]]
    iters = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    k = 27 * C
    kpad = -(-k // 64) * 64
    w = torch.zeros((N, kpad), dtype=BF, device="cuda")
    w[:, :k] = (torch.randn((N, k), device="cuda") / k ** 0.5).to(BF)
    b = torch.randn((N,), device="cuda").to(BF)
    resid = torch.randn((T * H * W, N), device="cuda").to(BF)
    src = (torch.randn((T + 2, H, W, C), device="cuda") * 0.7).to(BF)
    if C in ops.PADDED_CONV_CHANNELS:
        buf, hist, cur = ops.padded_activation(T, H, W, C, "cuda")
        hist.copy_(src[:2])
        cur.copy_(src[2:])
        fn = lambda: ops.vae_conv3d_padded(buf, w, b, resid=resid)      # noqa: E731
    else:
        fn = lambda: ops.vae_conv3d(src[2:], None, w, b, 3, 3, resid=resid, history_in_front=True)      # noqa: E731
    for _ in range(iters + 1):
        fn()
    torch.cuda.synchronize()
    print(f"conv {sys.argv[1]}: {C}->{N} over {T}x{H}x{W}, {iters + 1} launches, {2.0 * T * H * W * k * N / 1e12:.2f} TFLOP each")


if __name__ == "__main__":
    main()
