#!/usr/bin/env python3
"""profiles/pmc_static.json — the static HBM-side traffic figures bench.py prints in `roofline.traffic` / `roofline_gemm[*].traffic`.

    python3 tools/pmc_static.py <dir with the *_FETCH_SIZE.md / *_WRITE_SIZE.md tables of tools/profile_r05.sh> [<dir to cite>]

bytes per launch = (2 x FETCH_SIZE + WRITE_SIZE) KiB x 1024: on gfx950 FETCH_SIZE reports half of the bytes of a 16-B-per-lane
streaming read (MI355X_MICROARCH.md, HBM), WRITE_SIZE is exact for 16-B-per-lane stores.  Every entry records the sha256 of the
kernel sources it was measured on; bench.py withholds an entry whose sources have changed since (bench.static_traffic)."""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

ATTN_SRC = ["gf_attention.hip", "gf_common.h"]
GEMM_SRC = ["gf_gemm.hip", "gf_gemm_a4_loop.inc", "gf_gemm_a4f8_loop.inc", "gf_common.h"]
CONV_A4_SRC = ["gf_conv_a4.hip", "gf_conv_a4_loop.inc", "gf_common.h"]
CONV_C96_SRC = ["gf_conv_direct.hip", "gf_common.h"]
ENTRIES = {   # key -> (table tag, kernel-name substring, sources)
    "attn_self": ("attn", "flash_attn_fwd_kernel3", ATTN_SRC),
    "vae_conv_c192_n192_k33_m3": ("vae_l2", "conv_a4_kernel", CONV_A4_SRC),
    "vae_conv_c384_n384_k33_m3": ("vae_l1", "conv_a4_kernel", CONV_A4_SRC),
    "vae_conv_c96_n96_k33_m0": ("vae_c96", "conv3d_c96_kernel", CONV_C96_SRC),
    **{f"gemm_{sh}": (f"gemm_{sh}", "gemm_a4_kernel", GEMM_SRC) for sh in ("dd", "ffn1", "ffn2")},
    **{f"gemm_fp8_{sh}": (f"gemm_fp8_{sh}", "gemm_a4_kernel", GEMM_SRC) for sh in ("dd", "ffn1", "ffn2")},
}


def table_value(path, kernel_sub, counter):
    """mean sum per dispatch of `counter` for the kernel whose name contains `kernel_sub` (the row with most dispatches)."""
    best = None
    for line in open(path):
        cells = [c.strip() for c in line.strip().strip("|").split("|")]
        if len(cells) >= 6 and kernel_sub in cells[0] and cells[1] == counter:
            n, v = int(cells[2]), float(cells[4])
            if best is None or n > best[0]:
                best = (n, v)
    if best is None:
        raise KeyError(f"{path}: no row for {kernel_sub} / {counter}")
    return best[1]


def main():
    from bench import source_sha256
    src_dir = sys.argv[1]
    cite = sys.argv[2] if len(sys.argv) > 2 else src_dir
    out = {"entries": {}}
    try:
        from goal_force_amd import _lib
        out["gf_version"] = f"{_lib.version()} (C ABI {_lib.ABI_VERSION})"
    except Exception as e:      # noqa: BLE001 — the table is still valid without the version string
        out["gf_version"] = f"unknown ({type(e).__name__})"
    for key, (tag, sub, srcs) in ENTRIES.items():
        f, w = os.path.join(src_dir, f"{tag}_FETCH_SIZE.md"), os.path.join(src_dir, f"{tag}_WRITE_SIZE.md")
        try:
            fetch, write = table_value(f, sub, "FETCH_SIZE"), table_value(w, sub, "WRITE_SIZE")
        except (OSError, KeyError) as e:
            print(f"skip {key}: {e}", file=sys.stderr)
            continue
        out["entries"][key] = {"bytes_per_launch": int((2 * fetch + write) * 1024), "fetch_size_kib": fetch, "write_size_kib": write,
                               "from": f"{cite}/{tag}_FETCH_SIZE.md + {tag}_WRITE_SIZE.md (tools/profile_r05.sh)",
                               "sources": srcs, "sources_sha256": source_sha256(srcs)}
        print(f"{key}: {out['entries'][key]['bytes_per_launch'] / 1e9:.3f} GB per launch")
    with open(os.path.join(ROOT, "profiles", "pmc_static.json"), "w") as fh:
        json.dump(out, fh, indent=1)
    return 0 if out["entries"] else 1


if __name__ == "__main__":
    sys.exit(main())
