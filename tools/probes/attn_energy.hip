// Per-instruction ENERGY of the attention loop's ingredients under the package power limit (round 4).
// A v_mfma_f32_16x16x32_bf16 stream on random operands (2 waves per SIMD, every CU) with, per 32 MFMAs and wave, a chosen number of
//   ds_read_b128 operand reads from a random LDS image (kernel 3: 16; 64 query rows per wave: 8),
//   v_exp_f32 (16), v_cvt_pk_bf16_f32 (8), v_max3_f32 (8: kernel 3's running-maximum watch) or v_or3_b32 (4: GF_K3_ORMAX),
// each pinned as inline asm.  No barriers, no staging, no memory dependences.  One variant runs for a given number of seconds so
// that tools/energy_probe.py can read package power and clock beside it (rocm-smi); with the chip AT its power limit,
// energy per MFMA slot = power x time / slots, and differences between variants are the energy of the added instructions.
//   hipcc --offload-arch=gfx950 -O3 -o build/attn_energy tools/probes/attn_energy.hip;  ./build/attn_energy <variant> <seconds> [zeros|random] [1 = one wave per SIMD]
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

// per 32 MFMAs: RD reads, EX exp2, CV packs, MX max3, OR or3
template <int RD, int EX, int CV, int MX, int OR>
__global__ __launch_bounds__(512, 2) void k(const bf16x8* __restrict__ src, float* __restrict__ out, int iters) {
    __shared__ bf16x8 img[4096];                                   // 64 KiB
    const int t = threadIdx.x + blockIdx.x * blockDim.x;
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) img[i] = src[(i * 7 + blockIdx.x) & 0xffff];
    __syncthreads();
    bf16x8 a[4], b[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        a[i] = src[(t * 8 + i) & 0xffff];
        b[i] = src[(t * 8 + 4 + i) & 0xffff];
    }
    float x[16], p[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        x[i] = -0.01f * (float)((t * 16 + i) & 255);               // exponent arguments in (-2.56, 0]: P in (0.17, 1]
        p[i] = 0.5f;
    }
    float mx = -1e30f;
    unsigned orv = 0, w[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    f32x4 c[16] = {};
    // reads at cur + 256 k (k < 16): immediate offsets, conflict-free, no address VALU per read; the base alternates between two
    // images per iteration (one v_cndmask per 32 MFMAs) so that the loads are not loop-invariant
    typedef __attribute__((address_space(3))) const bf16x8* lds_ptr;
    lds_ptr rd0 = (lds_ptr)img + (threadIdx.x & 255);
    lds_ptr rd1 = (lds_ptr)img + ((threadIdx.x + 64) & 255);
    asm volatile("" : "+v"(rd0), "+v"(rd1));
    for (int it = 0; it < iters; ++it) {
        lds_ptr rd = (it & 1) ? rd1 : rd0;
#pragma unroll
        for (int n = 0; n < 32; ++n) {
            if constexpr (RD > 0) {
                if (n % (32 / RD) == 0) {
                    a[(n / (32 / RD)) & 3] = rd[256 * (n / (32 / RD))];          // 16 distinct 4 KiB strides inside the 64 KiB image
                }
            }
            if constexpr (EX > 0) {
                if (n % (32 / EX) == 0) {
                    const int e = (n / (32 / EX)) & 15;
                    asm volatile("v_exp_f32 %0, %1" : "=v"(p[e]) : "v"(x[e]));
                }
            }
            if constexpr (CV > 0) {
                if (n % (32 / CV) == 1) {
                    const int e = (n / (32 / CV)) & 7;
                    asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(w[e]) : "v"(p[2 * e]), "v"(p[2 * e + 1]));
                    u32x4 bb = __builtin_bit_cast(u32x4, b[e >> 2]);
                    bb[e & 3] = w[e];                                       // P feeds the next MFMAs' B operand, as in the kernel
                    b[e >> 2] = __builtin_bit_cast(bf16x8, bb);
                }
            }
            if constexpr (MX > 0) {
                if (n % (32 / MX) == 2) {
                    const int e = (n / (32 / MX)) & 7;
                    asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(mx) : "v"(x[2 * e]), "v"(x[2 * e + 1]));
                }
            }
            if constexpr (OR > 0) {
                if (n % (32 / OR) == 3) {
                    const int e = (n / (32 / OR)) & 3;
                    const u32x4 q = __builtin_bit_cast(u32x4, b[e >> 1]);          // the packed words where they already live
                    asm volatile("v_or3_b32 %0, %0, %1, %2" : "+v"(orv) : "v"(q[2 * (e & 1)]), "v"(q[2 * (e & 1) + 1]));
                }
            }
            c[n & 15] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[n & 3], b[(n >> 2) & 3], c[n & 15], 0, 0, 0);
        }
    }
    float s = mx + (float)(orv & 1);
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        s += p[j];
#pragma unroll
        for (int e = 0; e < 4; ++e) s += c[j][e];
    }
    out[t] = s;
}

struct Variant {
    const char* name;
    void (*fn)(const bf16x8*, float*, int);
    int rd, ex, cv, mx, orr;
};
#define V(name, RD, EX, CV, MX, OR) {name, k<RD, EX, CV, MX, OR>, RD, EX, CV, MX, OR}
static const Variant variants[] = {
    V("bare MFMA 16x16x32", 0, 0, 0, 0, 0),
    V("+ ds_read_b128 1 per 2 MFMAs (kernel 3)", 16, 0, 0, 0, 0),
    V("+ ds_read_b128 1 per 4 MFMAs (64 rows per wave)", 8, 0, 0, 0, 0),
    V("+ 16 v_exp_f32", 0, 16, 0, 0, 0),
    V("+ 16 v_exp_f32 + 8 v_cvt_pk_bf16_f32", 0, 16, 8, 0, 0),
    V("+ exp + cvt + 8 v_max3_f32 (kernel 3's VALU)", 0, 16, 8, 8, 0),
    V("+ exp + cvt + 4 v_or3_b32 (ORMAX's VALU)", 0, 16, 8, 0, 4),
    V("kernel 3's mix: VALU (max3) + 1 read per 2", 16, 16, 8, 8, 0),
    V("ORMAX's mix: VALU (or3) + 1 read per 2", 16, 16, 8, 0, 4),
    V("64-rows-per-wave mix: VALU (max3) + 1 read per 4", 8, 16, 8, 8, 0),
};

int main(int argc, char** argv) {
    const int nv = (int)(sizeof(variants) / sizeof(variants[0]));
    if (argc < 3) {
        for (int i = 0; i < nv; ++i) printf("%d %s\n", i, variants[i].name);
        return 0;
    }
    const int v = atoi(argv[1]);
    const double seconds = atof(argv[2]);
    const bool zeros = argc > 3 && !strcmp(argv[3], "zeros");
    const bool one_wave = argc > 4 && !strcmp(argv[4], "1");      // 256-thread workgroups: ONE wave per SIMD
    if (v < 0 || v >= nv) return 2;
    const int n = 1 << 16;
    std::vector<unsigned short> h(n * 8);
    srand(1);
    bf16x8* d;
    float* out;
    hipMalloc(&d, n * 16);
    hipMalloc(&out, 1024 * 512 * 4);
    for (auto& x : h) {
        const float f = ((rand() / (float)RAND_MAX) + (rand() / (float)RAND_MAX) + (rand() / (float)RAND_MAX) - 1.5f) * 2.f;
        unsigned u;
        memcpy(&u, &f, 4);
        x = (unsigned short)(u >> 16);
    }
    if (zeros) hipMemset(d, 0, n * 16);
    else hipMemcpy(d, h.data(), n * 16, hipMemcpyHostToDevice);
    const int iters = 20000, blocks = 256, threads = one_wave ? 256 : 512;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(variants[v].fn, dim3(blocks), dim3(threads), 0, 0, d, out, iters);      // warm-up
    hipDeviceSynchronize();
    const auto t0 = std::chrono::steady_clock::now();
    int launches = 0;
    hipEventRecord(e0);
    while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < seconds) {
        for (int i = 0; i < 8; ++i) hipLaunchKernelGGL(variants[v].fn, dim3(blocks), dim3(threads), 0, 0, d, out, iters);
        launches += 8;
        hipStreamSynchronize(0);
    }
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double slots = (double)launches * blocks * (threads / 64) * iters * 32.0;       // wave-level MFMA slots executed
    printf("{\"variant\": %d, \"waves_per_simd\": %d, \"name\": \"%s\", \"data\": \"%s\", \"seconds\": %.3f, \"mfma_slots\": %.6e, \"tflops\": %.1f, "
           "\"per32\": {\"ds_read_b128\": %d, \"v_exp_f32\": %d, \"v_cvt_pk_bf16_f32\": %d, \"v_max3_f32\": %d, \"v_or3_b32\": %d}}\n",
           v, one_wave ? 1 : 2, variants[v].name, zeros ? "zeros" : "random", ms / 1e3, slots, slots * 16384.0 / (ms * 1e-3) / 1e12, variants[v].rd,
           variants[v].ex, variants[v].cv, variants[v].mx, variants[v].orr);
    return 0;
}
