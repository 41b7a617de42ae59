// TIMING-ONLY probe: 64 query rows per wave, one wave per SIMD, hand-scheduled asm phase (tools/gen_attn_nq4.py) — see that file.
//   python3 tools/gen_attn_nq4.py > build/attn_nq4_loop.inc
//   hipcc --offload-arch=gfx950 -O3 -Ibuild -o build/attn_nq4_whatif tools/probes/attn_nq4_whatif.hip;  ./build/attn_nq4_whatif [zeros] [tiles]
// Same problem as the shipped self-attention: S = 32760 queries and keys, 40 heads x 128; a workgroup (4 waves) owns 256 queries of one
// head and walks all 512 key tiles; K [S, 5120] and V^T [40][128][kv_pad] are staged by LDS-DMA exactly as kernel 3 stages them.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "attn_nq4_loop.inc"
typedef unsigned short u16;

template <int I>
__device__ __forceinline__ float acc_read() {
    float x;
    asm volatile("v_accvgpr_read_b32 %0, a[%1]" : "=v"(x) : "n"(I));
    return x;
}

__global__ __launch_bounds__(256, 1) void nq4_kernel(const u16* q, const u16* k, const u16* vt, float* out, int kv_len, int heads, int n_qblocks,
                                                     long q_stride, long k_stride, long kv_pad, int tiles) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 15, g = lane >> 4;
    const int pid = blockIdx.x, xcd = pid & 7, idx = pid >> 3;
    const int head = xcd + 8 * (idx / n_qblocks), qb0 = idx % n_qblocks;
    const int q0 = qb0 * 256 + wave * 64;
    const unsigned lbase = (unsigned)(unsigned long)(__attribute__((address_space(3))) char*)smem;
    // LDS fragment read addresses (kernel 3's images): K (kb, ks): koff[ks] + 4096 kb;  V^T (db, kk): voff[kk] + 2048 db
    unsigned koff0 = lbase + 256 * r + 16 * ((0 + g) ^ r), koff1 = lbase + 256 * r + 16 * ((4 + g) ^ r);
    unsigned koff2 = lbase + 256 * r + 16 * ((8 + g) ^ r), koff3 = lbase + 256 * r + 16 * ((12 + g) ^ r);
    unsigned voff0 = lbase + 32768 + 128 * r + 16 * ((0 + g) ^ ((r >> 1) & 7)), voff1 = lbase + 32768 + 128 * r + 16 * ((4 + g) ^ ((r >> 1) & 7));
    // staging: piece 4 jj + wave of a tile; lane offsets as in kernel 3 (swizzle independent of jj)
    const int row = 4 * wave + (lane >> 4), lch = (lane & 15) ^ (row & 15);
    unsigned ksoff = ((unsigned)row * (unsigned)k_stride + head * 128 + lch * 8) * 2u;
    const int vrow = 8 * wave + (lane >> 3), vch = (lane & 7) ^ ((vrow >> 1) & 7);
    unsigned vtoff = (unsigned)(((long)head * 128 + vrow) * kv_pad + vch * 8) * 2u;
    unsigned qoff = (unsigned)(((long)min(q0 + r, kv_len - 1) * q_stride + head * 128 + 8 * g) * 2);
    const unsigned qstep = (unsigned)(16 * q_stride * 2);
    const unsigned long kb = (unsigned long)k, vb = (unsigned long)vt;
    const unsigned kLo = (unsigned)kb, kHi = (unsigned)(kb >> 32) & 0xffffu, kNr = (unsigned)((long)kv_len * k_stride * 2);
    const unsigned vLo = (unsigned)vb, vHi = (unsigned)(vb >> 32) & 0xffffu, vNr = (unsigned)((long)heads * 128 * kv_pad * 2);
    const unsigned ldsW = lbase + (unsigned)wave * 1024u;
    const unsigned npairs = (unsigned)(tiles / 2);
    const unsigned kstep = (unsigned)(64 * k_stride * 2), kpiece = (unsigned)(16 * k_stride * 2), vpiece = (unsigned)(32 * kv_pad * 2);
    const u16* qptr = q;
    const unsigned wdst = ldsW + (unsigned)lane * 16u;       // register-staged variant: this lane's 16 bytes of a 1 KiB piece
    GF_NQ4_LOOP_ASM(koff0, koff1, koff2, koff3, voff0, voff1, ksoff, vtoff, qoff, qstep, qptr, kLo, kHi, kNr, vLo, vHi, vNr, ldsW, npairs, kstep,
                    kpiece, vpiece, wdst);
    // keep every accumulator live: a checksum per lane (results are meaningless by construction)
    float s = acc_read<0>() + acc_read<37>() + acc_read<70>() + acc_read<101>() + acc_read<127>() + acc_read<128>() + acc_read<143>();
    out[(long)blockIdx.x * 256 + tid] = s;
}

int main(int argc, char** argv) {
    const bool zeros = argc > 1 && !strcmp(argv[1], "zeros");
    const int S = 32760, H = 40, D = 5120;
    const int tiles = argc > 2 ? atoi(argv[2]) : 512;
    const long kv_pad = (S + 63) / 64 * 64;
    const int nqb = (S + 255) / 256;
    std::vector<u16> h((size_t)S * D);
    srand(1);
    auto fill = [&](float scale) {
        for (auto& x : h) {
            const float f = zeros ? 0.f : ((rand() / (float)RAND_MAX) + (rand() / (float)RAND_MAX) + (rand() / (float)RAND_MAX) - 1.5f) * 2.f * scale;
            unsigned u;
            memcpy(&u, &f, 4);
            x = (u16)(u >> 16);
        }
    };
    u16 *q, *k, *vt;
    float* out;
    hipMalloc(&q, (size_t)S * D * 2);
    hipMalloc(&k, (size_t)(S + 256) * D * 2);          // the loop requests K tiles nt, nt + 1 through the SGPR offset, which the
    hipMemset(k, 0, (size_t)(S + 256) * D * 2);        // descriptor does not bounds-check: those rows exist and are zero
    hipMalloc(&vt, (size_t)H * 128 * kv_pad * 2);
    hipMalloc(&out, (size_t)nqb * H * 256 * 4);
    fill(0.1275f);                        // Q carries scale * log2(e), as in the shipped kernel
    hipMemcpy(q, h.data(), (size_t)S * D * 2, hipMemcpyHostToDevice);
    fill(1.f);
    hipMemcpy(k, h.data(), (size_t)S * D * 2, hipMemcpyHostToDevice);
    hipMemcpy(vt, h.data(), (size_t)H * 128 * kv_pad * 2 <= (size_t)S * D * 2 ? (size_t)H * 128 * kv_pad * 2 : (size_t)S * D * 2, hipMemcpyHostToDevice);
    hipFuncSetAttribute(reinterpret_cast<const void*>(nq4_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    float best = 1e9f;
    for (int rep = 0; rep < 6; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(nq4_kernel, dim3(nqb * H), dim3(256), 65536, 0, q, k, vt, out, S, H, nqb, (long)D, (long)D, kv_pad, tiles);
        hipEventRecord(e1);
        if (hipEventSynchronize(e1) != hipSuccess) {
            printf("launch failed: %s\n", hipGetErrorString(hipGetLastError()));
            return 1;
        }
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (rep >= 2 && ms < best) best = ms;
    }
    const double alg = 4.0 * S * (double)(tiles * 64) * D;
    printf("nq4 timing-only loop, %s data, %d tiles: %.3f ms = %.1f TFLOP/s algorithmic (%.1f executed, 136 MFMAs per 128)\n", zeros ? "zero" : "random",
           tiles, best, alg / best / 1e9, alg * 136.0 / 128.0 / best / 1e9);
    return 0;
}
