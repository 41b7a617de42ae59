import torch
x = torch.randn((32760, 5120), device="cuda").to(torch.bfloat16)
y = torch.empty_like(x)
for name, fn in (("copy_", lambda: y.copy_(x)), ("add_", lambda: torch.add(x, x, out=y)), ("mul scalar", lambda: torch.mul(x, 2.0, out=y))):
    fn(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): fn()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 20)
    print(f"{name}: {best:.3f} ms = {2 * x.numel() * 2 / best / 1e9:.2f} TB/s (read+write 671 MB)")
