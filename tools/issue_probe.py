#!/usr/bin/env python3
"""Issue-rate probe for gfx950: how many cycles does one 32x32x16 bf16 MFMA "slot" take when VALU / transcendental /
LDS instructions of the same wave (1 wave per SIMD) or of a SIMD partner (2 waves per SIMD) are issued beside it?
Generates a HIP file whose kernels are straight inline-asm loops (no compiler scheduling involved), builds it with
hipcc and, with --run, times every mix.  The flash-attention slot schedule (gf_attention.hip kernel 2) is sized from
these numbers; results are kept in profiles/.

  python3 tools/issue_probe.py --build          # here (cross-compiles)
  python3 tools/issue_probe.py --run            # on the GPU box
"""
import argparse, os, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "build", "issue_probe.hip")
BIN = os.path.join(ROOT, "build", "issue_probe")

# one slot = one MFMA (or none) + extras; 8 slots per loop trip, accumulators rotate over a[0:63]
MIXES = [
    ("mfma", dict(m=1)),
    ("mfma+1exp", dict(m=1, exp=1)),
    ("mfma+2exp", dict(m=1, exp=2)),
    ("mfma+4exp", dict(m=1, exp=4)),
    ("mfma+2fma", dict(m=1, fma=2)),
    ("mfma+4fma", dict(m=1, fma=4)),
    ("mfma+6fma", dict(m=1, fma=6)),
    ("mfma+8fma", dict(m=1, fma=8)),
    ("mfma+4salu", dict(m=1, salu=4)),
    ("mfma+1exp+3fma", dict(m=1, exp=1, fma=3)),
    ("mfma+1exp+3fma+2lds", dict(m=1, exp=1, fma=3, lds=2)),
    ("mfma+1exp+3fma+2lds+2salu", dict(m=1, exp=1, fma=3, lds=2, salu=2)),
    ("mfma+2lds", dict(m=1, lds=2)),
    ("exp", dict(exp=1)),
    ("fma", dict(fma=1)),
    ("pkfma", dict(pk=1)),
    ("lds_tr", dict(lds=1)),
    ("lds_b128", dict(ldsq=1)),
    ("lds_b128_lin16", dict(ldsq=1, addr="lin16")),
    ("lds_b128_kpattern", dict(ldsq=1, addr="kpat")),
    ("lds_tr_vpattern", dict(lds=1, addr="vpat")),
    ("mfma+1lds_b128_k", dict(m=1, ldsq=1, addr="kpat")),
    ("mfma+2lds_tr_v", dict(m=1, lds=2, addr="vpat")),
    ("mfmaV (acc in VGPRs)", dict(mv=1)),
    ("mfmaV+2lds_tr_v", dict(mv=1, lds=2, addr="vpat")),
    ("mfmaV+1exp+3fma", dict(mv=1, exp=1, fma=3)),
    ("mfmaV+1exp+3fma+2lds", dict(mv=1, exp=1, fma=3, lds=2, addr="vpat")),
    ("mfma+1exp+3fma+2lds_v", dict(m=1, exp=1, fma=3, lds=2, addr="vpat")),
    ("mfma+1exp+2fma+1cvt+2lds+wait", dict(m=1, exp=1, fma=2, cvt=1, lds=2, addr="vpat", wait=1)),
    ("mfmaV+1exp+2fma+1cvt+2lds+wait", dict(mv=1, exp=1, fma=2, cvt=1, lds=2, addr="vpat", wait=1)),
    ("lds_b128 gemm frag 16 rows x 4 chunks", dict(ldsq=1, addr="g16")),
    ("lds_b128 gemm frag 32 rows x 2 chunks", dict(ldsq=1, addr="g32")),
    ("lds_b128 gemm frag 32 rows, pair swizzle", dict(ldsq=1, addr="g32s")),
    ("lds_b128 gemm frag 32 rows, pair swizzle ks=1", dict(ldsq=1, addr="g32s1")),
    ("lds_b128 gemm frag 32 rows, swz (row>>1)&7", dict(ldsq=1, addr="g32t")),
    ("mfma 1 accumulator chain", dict(m=1, nacc=1)),
    ("mfma 2 accumulator chains", dict(m=1, nacc=2)),
    ("GEMM-X: 8 mfma32 + 4 ldsq + 2 dma / trip", dict(m=1, ldsq_every=2, dma_trip=2, addr="lin16")),
    ("GEMM-X: 8 mfma32 + 4 ldsq + 1 dma / trip", dict(m=1, ldsq_every=2, dma_trip=1, addr="lin16")),
    ("GEMM-X: 8 mfma32 + 4 ldsq (no dma)", dict(m=1, ldsq_every=2, addr="lin16")),
    ("GEMM-X: 8 mfma32 + 2 dma (no lds)", dict(m=1, dma_trip=2, addr="lin16")),
    ("GEMM-ph: 8 mfma16 + 3 ldsq + 1 dma / trip", dict(m16=1, ldsq_trip=3, dma_trip=1, addr="lin16")),
    ("dma only (2 / trip)", dict(dma_trip=2, addr="lin16")),
    ("dma only, 8 rows x 128 B", dict(dma_trip=2, addr="lin16", gaddr="r8")),
    ("dma only, 16 rows x 64 B", dict(dma_trip=2, addr="lin16", gaddr="r16")),
    ("GEMM-X mix, dma 8 rows x 128 B", dict(m=1, ldsq_every=2, dma_trip=2, addr="lin16", gaddr="r8")),
    ("GEMM-X mix, dma 16 rows x 64 B", dict(m=1, ldsq_every=2, dma_trip=2, addr="lin16", gaddr="r16")),
    ("mfma16", dict(m16=1)),
    ("mfma+2pk+1exp", dict(m=1, pk=2, exp=1)),
]


def body(mix):
    lines = []
    for s in range(8):
        a0 = 16 * (s % mix.get("nacc", 4))
        if mix.get("m"):
            lines.append(f"v_mfma_f32_32x32x16_bf16 a[{a0}:{a0+15}], v[8:11], v[12:15], a[{a0}:{a0+15}]")
        if mix.get("mv"):
            v0 = 120 + 16 * (s % 4)
            lines.append(f"v_mfma_f32_32x32x16_bf16 v[{v0}:{v0+15}], v[8:11], v[12:15], v[{v0}:{v0+15}]")
        if mix.get("m16"):
            a4 = 4 * (s % 8)
            lines.append(f"v_mfma_f32_16x16x32_bf16 a[{a4}:{a4+3}], v[8:11], v[12:15], a[{a4}:{a4+3}]")
        for k in range(mix.get("lds", 0)):
            r = 52 + 2 * ((2 * s + k) % 16)
            lines.append(f"ds_read_b64_tr_b16 v[{r}:{r+1}], v16 offset:{4096 * ((2 * s + k) % 4)}")
        nq = mix.get("ldsq", 0)
        if mix.get("ldsq_every") and s % mix["ldsq_every"] == 0:
            nq = 1
        if mix.get("ldsq_trip") and s < mix["ldsq_trip"]:
            nq = 1
        if mix.get("dma_trip") and s in ((1, 5)[:mix["dma_trip"]]):
            lines.append(f"s_mov_b32 m0, {32768 + 1024 * s}")
            lines.append("s_nop 0")
            lines.append("global_load_lds_dwordx4 v[2:3], off")
        for k in range(nq):
            r = 52 + 4 * (s % 8)
            lines.append(f"ds_read_b128 v[{r}:{r+3}], v16 offset:{8192 * (s % 4)}")
        for k in range(mix.get("exp", 0)):
            lines.append(f"v_exp_f32 v{20 + (4 * s + k) % 32}, v17")
        for k in range(mix.get("fma", 0)):
            lines.append(f"v_fma_f32 v{84 + (8 * s + k) % 32}, v17, v18, v19")
        for k in range(mix.get("pk", 0)):
            r = 84 + 2 * ((8 * s + k) % 16)
            lines.append(f"v_pk_fma_f32 v[{r}:{r+1}], v[8:9], v[10:11], v[12:13]")
        for k in range(mix.get("cvt", 0)):
            lines.append(f"v_cvt_pk_bf16_f32 v{116 + (s % 4)}, v17, v18")
        if mix.get("wait"):
            lines.append("s_waitcnt lgkmcnt(6)")
        for k in range(mix.get("salu", 0)):
            lines.append(f"s_add_u32 s{24 + k}, s{24 + k}, 1")
        if mix.get("lds") or mix.get("ldsq") or mix.get("ldsq_every") or mix.get("ldsq_trip"):
            if s % 4 == 3:
                lines.append("s_waitcnt lgkmcnt(4)")
        if mix.get("dma_trip") and s == 7:
            lines.append("s_waitcnt vmcnt(4)")
    return lines


def gen():
    out = ['#include <hip/hip_runtime.h>', '#include <cstdio>', '#include <cstdlib>', '#include <cstring>',
           'extern __shared__ char smem[];']
    clob = ",".join([f'"v{i}"' for i in range(8, 184)] + [f'"a{i}"' for i in range(64)] +
                    [f'"s{i}"' for i in range(20, 30)] + ['"v2"', '"v3"', '"m0"', '"scc"', '"memory"'])
    for idx, (name, mix) in enumerate(MIXES):
        asm = ["v_mov_b32 v16, %1", "v_mov_b32 v17, 0x3f000000", "v_mov_b32 v18, 0x3f800000", "v_mov_b32 v19, 0",
               "s_mov_b32 s20, %0", "v_mov_b32 v2, %2", "v_mov_b32 v3, %3"]
        asm += [f"v_mov_b32 v{r}, 0" for r in range(8, 16)]
        asm += [f"v_accvgpr_write_b32 a{r}, 0" for r in range(64)]
        asm += ["s_nop 7", "1:"] + body(mix) + ["s_sub_u32 s20, s20, 1", "s_cmp_lg_u32 s20, 0", "s_cbranch_scc1 1b",
                                              "s_waitcnt lgkmcnt(0)", "s_nop 15"]
        text = "\\n\\t".join(asm)
        addr = {"lin8": "lane * 8", "lin16": "lane * 16",
                "kpat": "256 * (lane & 31) + 16 * ((lane >> 5) ^ (((lane & 3) << 2) | ((lane >> 2) & 3)))",
                "g16": "128 * (lane & 15) + 16 * ((lane >> 4) ^ (lane & 7))",
                "g32": "128 * (lane & 31) + 16 * ((lane >> 5) ^ (lane & 7))",
                "g32s": "256 * ((lane & 31) >> 1) + 16 * ((((lane & 1) * 8) + (lane >> 5)) ^ ((lane & 31) >> 1))",
                "g32s1": "256 * ((lane & 31) >> 1) + 16 * ((((lane & 1) * 8) + 2 + (lane >> 5)) ^ ((lane & 31) >> 1))",
                "g32t": "128 * (lane & 31) + 16 * ((lane >> 5) ^ ((lane >> 1) & 7))",
                "vpat": "256 * (4 * (lane >> 5) + ((lane & 15) >> 2)) + 16 * ((2 * ((lane >> 4) & 1) + ((lane & 3) >> 1)) ^ ((((lane & 15) >> 2) << 2) | (lane >> 5))) + 8 * (lane & 1)",
                }[mix.get("addr", "lin8")]
        gexpr = {"lin": "lane * 16", "r8": "(lane >> 3) * 10240 + (lane & 7) * 16",
                 "r16": "(lane >> 2) * 10240 + (lane & 3) * 16"}[mix.get("gaddr", "lin")]
        out.append(f'__global__ void probe{idx}(int n, float* o) {{\n'
                   f'  int lane = threadIdx.x & 63;\n'
                   f'  int off = {addr};\n'
                   f'  unsigned long long ga = (unsigned long long)(o + (blockIdx.x & 255) * 4096) + {gexpr};\n'
                   f'  unsigned galo = (unsigned)ga, gahi = (unsigned)(ga >> 32);\n'
                   f'  asm volatile("{text}\\n\\ts_waitcnt vmcnt(0)" :: "s"(n), "v"(off), "v"(galo), "v"(gahi) : {clob});\n'
                   f'  if (n < 0) o[threadIdx.x] = 0.f;\n}}')
    out.append("typedef void (*kfn)(int, float*);")
    out.append("static kfn fns[] = {" + ",".join(f"probe{i}" for i in range(len(MIXES))) + "};")
    out.append("static const char* names[] = {" + ",".join(f'"{n}"' for n, _ in MIXES) + "};")
    out.append(r'''
int main() {
  const int n = 20000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float* gbuf = nullptr; if (hipMalloc(&gbuf, 64 << 20) != hipSuccess) { printf("malloc failed\n"); return 1; }
  hipMemset(gbuf, 0, 64 << 20);
  for (int wps = 1; wps <= 2; ++wps) {
    double base = 0;
    for (unsigned i = 0; i < sizeof(fns) / sizeof(fns[0]); ++i) {
      hipFuncSetAttribute((const void*)fns[i], hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
      float best = 1e30f;
      printf("waves/SIMD=%d %-28s ", wps, names[i]); fflush(stdout);
      for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(fns[i], dim3(256), dim3(256 * wps), 65536, 0, n, gbuf);
        hipEventRecord(e1, 0);
        if (hipEventSynchronize(e1) != hipSuccess) { printf("sync error\n"); return 1; }
        float ms; hipEventElapsedTime(&ms, e0, e1); if (rep && ms < best) best = ms;
      }
      if (hipGetLastError() != hipSuccess) { printf("launch error\n"); return 1; }
      double ns_slot = best * 1e6 / (n * 8.0);
      if (i == 0) base = ns_slot;   // one MFMA = 8 passes = 32 cycles
      printf("%8.2f ns/slot  = %6.1f cycles (clock from mfma-only = %.2f GHz)\n",
             ns_slot, ns_slot / base * 32.0, 32.0 / base);
    }
  }
  return 0;
}''')
    os.makedirs(os.path.dirname(SRC), exist_ok=True)
    open(SRC, "w").write("\n".join(out) + "\n")


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--build", action="store_true")
    ap.add_argument("--run", action="store_true")
    a = ap.parse_args()
    if a.build:
        gen()
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O2", "-o", BIN, SRC])
    if a.run:
        sys.exit(subprocess.call([BIN]))
