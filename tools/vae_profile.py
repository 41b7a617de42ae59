#!/usr/bin/env python3
"""Run the tiled VAE decode (and optionally encode) of one 81-frame 480x832 video once or twice, for
`rocprofv3 --kernel-trace --stats -- python3 tools/vae_profile.py [decode|encode] [reps]` (random-init weights)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from goal_force_amd.vae import WanVideoVAE  # noqa: E402


def main():
    what = sys.argv[1] if len(sys.argv) > 1 else "decode"
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    dev = torch.device("cuda", 0)
    torch.set_grad_enabled(False)
    torch.manual_seed(7)
    vae = WanVideoVAE().to(torch.bfloat16).to(dev)
    kw = dict(device=dev, tiled=True, tile_size=(30, 52), tile_stride=(15, 26))
    if what == "encode":
        x = [(torch.rand((3, 81, 480, 832), device=dev) * 2 - 1).to(torch.bfloat16)]
        fn = lambda: vae.encode(x, **kw)
    else:
        z = torch.randn((1, 16, 21, 60, 104), device=dev).to(torch.bfloat16)
        fn = lambda: vae.decode(z, **kw)
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        print(f"{what}: {time.perf_counter() - t0:.3f} s", flush=True)


if __name__ == "__main__":
    main()
