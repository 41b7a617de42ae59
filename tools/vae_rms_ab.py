import os, sys, torch
sys.path.insert(0, "/root/repo")
from goal_force_amd import ops
BF = torch.bfloat16
for shape in ((40, 240, 416, 96), (40, 120, 208, 192), (20, 60, 104, 384)):
    x = (torch.randn(shape, device="cuda") * 2).to(BF)
    gam = torch.ones(shape[-1], device="cuda").to(BF)
    out = torch.empty_like(x)
    res = {}
    for rnd in range(3):
        for name, env in (("pow2", "0"), ("three", "1")):
            with ops.options(vae_rms3=int(env)):
                ops.vae_rmsnorm_silu(x, gam, silu=True, out=out)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(5):
                    ops.vae_rmsnorm_silu(x, gam, silu=True, out=out)
                e1.record()
                torch.cuda.synchronize()
                res[name] = min(res.get(name, 1e9), e0.elapsed_time(e1) / 5)
    gb = 2 * x.numel() * 2 / 1e9
    print(shape, {k: f"{v:.3f} ms = {gb / v:.2f} TB/s" for k, v in res.items()}, flush=True)
