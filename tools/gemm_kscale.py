# NOTE (round 5): this script drives variants that are no longer in the product library (GF_* environment selectors, v1 / sl
# kernels, what-if builds).  It runs against a library built from the experimental tree: `bash tools/experimental_tree.sh`, then build
# build/experimental/csrc as the Makefile builds goal_force_amd/csrc and point GOALFORCE_HIP_LIB at the result.
import os, sys, torch
sys.path.insert(0, os.getcwd())
from goal_force_amd import ops
from tools.microbench import timeit
BF = torch.bfloat16
S = 32760
for N in (13824, 5120):
    for K in (640, 1280, 2560, 5120, 10240, 13824):
        x = torch.randn((S, K), device="cuda").to(BF)
        w = (torch.randn((N, K), device="cuda") / K ** 0.5).to(BF)
        out = torch.empty((S, N), device="cuda", dtype=BF)
        res = {}
        for kern in ("a4", "ph"):
            os.environ["GF_GEMM_KERNEL"] = kern
            ops.reload_options()       # the launch paths read the knobs once per process
            res[kern] = timeit(lambda: ops.gemm(x, w, None, out=out), 8)[0]
        tl = torch.nn.functional.linear
        ref = timeit(lambda: tl(x, w), 8)[0]
        fl = 2.0 * S * N * K
        tiles = 128 * ((N + 255) // 256) / 256
        print(f"N={N} K={K}: a4 {res['a4']:.3f} ms ({fl/res['a4']/1e9:.0f} TF) ph {res['ph']:.3f} ({fl/res['ph']/1e9:.0f}) blaslt {ref:.3f} ({fl/ref/1e9:.0f});  a4 per tile {res['a4']*1e3/tiles:.1f} us, per K-iter {res['a4']*1e6/tiles/(K/64):.0f} ns")
