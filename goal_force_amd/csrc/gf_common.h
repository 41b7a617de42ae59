// gf_common.h — shared device helpers for the goalforce HIP kernels (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "goalforce.h"

typedef unsigned short u16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(8))) unsigned short u16x8;
typedef __attribute__((ext_vector_type(4))) unsigned short u16x4;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

#define GF_LDS __attribute__((address_space(3)))
#define GF_GLOBAL __attribute__((address_space(1)))

// bf16 <-> f32.  The plain cast lowers to v_cvt_pk_bf16_f32 (round-to-nearest-even,
// NaN stays NaN) on gfx950.
__device__ __forceinline__ float bf2f(u16 u) { return __uint_as_float(((unsigned)u) << 16); }
__device__ __forceinline__ u16 f2bf(float f) { return __builtin_bit_cast(u16, (__bf16)f); }
// round an fp32 value to the nearest bf16 and return it as fp32 (the reference's
// eager bf16 ops round after every elementwise op; we reproduce those roundings)
__device__ __forceinline__ float rbf(float f) { return bf2f(f2bf(f)); }
// two fp32 -> one dword of two bf16 (lo in bits 0-15): the vector conversion lowers to ONE v_cvt_pk_bf16_f32 (round-to-nearest-even,
// the instruction the scalar casts use too); `f2bf(lo) | f2bf(hi) << 16` compiled to the convert plus an and / shift / or per pair
__device__ __forceinline__ unsigned pack2bf(float lo, float hi) {
    typedef __attribute__((ext_vector_type(2))) float f2_t;
    typedef __attribute__((ext_vector_type(2))) __bf16 b2_t;
    return __builtin_bit_cast(unsigned, __builtin_convertvector(f2_t{lo, hi}, b2_t));
}

__device__ __forceinline__ float gelu_tanh_f(float x) {
    // torch GELU(approximate='tanh'): 0.5*x*(1+tanh(sqrt(2/pi)*(x+0.044715*x^3)))
    // = x * sigmoid(2u), u = sqrt(2/pi) * (x + 0.044715 x^3): one exp2 and one rcp instead of the tanhf library call (a GEMM
    // epilogue runs this 128 times per lane and tile); fp32 error ~3 ulp, far inside the bf16 rounding that follows.
    const float a = -2.0f * 0.7978845608028654f * 1.4426950408889634f, b = a * 0.044715f;
    const float t = __builtin_amdgcn_exp2f(x * __builtin_fmaf(b, x * x, a));   // exp(-2u); inf for very negative x -> -0
    return x * __builtin_amdgcn_rcpf(1.0f + t);
}
// four at a time: the same operations as gelu_tanh_f, the multiplies / adds on packed fp32 pairs (v_pk_mul / v_pk_fma / v_pk_add),
// the four exp2 and the four rcp back to back so that no transcendental result is needed by the very next instruction —
// bit-identical to four gelu_tanh_f calls, about half the issue slots
typedef __attribute__((ext_vector_type(2))) float gf_f32x2;
__device__ __forceinline__ void gelu_tanh_f4(gf_f32x2& x01, gf_f32x2& x23) {
    const float a = -2.0f * 0.7978845608028654f * 1.4426950408889634f, b = a * 0.044715f;
    const gf_f32x2 av = {a, a}, bv = {b, b}, one = {1.0f, 1.0f};
    const gf_f32x2 g01 = x01 * __builtin_elementwise_fma(bv, x01 * x01, av), g23 = x23 * __builtin_elementwise_fma(bv, x23 * x23, av);
    const float t0 = __builtin_amdgcn_exp2f(g01[0]), t1 = __builtin_amdgcn_exp2f(g01[1]);
    const float t2 = __builtin_amdgcn_exp2f(g23[0]), t3 = __builtin_amdgcn_exp2f(g23[1]);
    const gf_f32x2 d01 = one + gf_f32x2{t0, t1}, d23 = one + gf_f32x2{t2, t3};
    const float r0 = __builtin_amdgcn_rcpf(d01[0]), r1 = __builtin_amdgcn_rcpf(d01[1]);
    const float r2 = __builtin_amdgcn_rcpf(d23[0]), r3 = __builtin_amdgcn_rcpf(d23[1]);
    x01 = x01 * gf_f32x2{r0, r1};
    x23 = x23 * gf_f32x2{r2, r3};
}
__device__ __forceinline__ float silu_f(float x) { return x / (1.0f + __expf(-x)); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// block-wide sum for blockDim.x == THREADS (multiple of 64); `red` is THREADS/64 floats of LDS
template <int THREADS>
__device__ __forceinline__ float block_sum(float v, float* red) {
    v = wave_sum(v);
    const int w = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[w] = v;
    __syncthreads();
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < THREADS / 64; ++i) t += red[i];
    return t;
}

// host side ---------------------------------------------------------------
void gf_set_error(const char* fmt, ...);
#define GF_CHECK_ARG(cond, ...)                 \
    do {                                        \
        if (!(cond)) {                          \
            gf_set_error(__VA_ARGS__);          \
            return GF_ERR_INVALID_ARG;          \
        }                                       \
    } while (0)
#define GF_CHECK_LAUNCH(name)                                                        \
    do {                                                                             \
        hipError_t e_ = hipGetLastError();                                           \
        if (e_ != hipSuccess) {                                                      \
            gf_set_error("%s: launch failed: %s", name, hipGetErrorString(e_));      \
            return GF_ERR_LAUNCH;                                                    \
        }                                                                            \
    } while (0)
static inline bool gf_aligned16(const void* p) { return (((uintptr_t)p) & 15u) == 0; }

// Dispatch overrides of the launchers.  Every kernel in this library ships: each serves the shapes its launcher sends it.  The
// parity tests cross-check two of them on the SAME operands (bit-identical by construction), which needs a way to send a shape to
// the kernel that would not get it by default: gf_set_option(name, value) (gf_abi.hip).  The library reads NO environment variable;
// the defaults below are the shipped dispatch, and nothing but an explicit call changes them.  Launch paths do relaxed loads.
#include <atomic>
struct GfOptions {
    std::atomic<int> prefer_8wave{0};   // "prefer_8wave": 1 = GEMMs with M >= 512 on the 8-wave kernels (default: the 4-wave kernel)
    std::atomic<int> a4_stagger{2};     // "a4_stagger": K tiles between the K-loop starts of neighbouring column tiles (>= 0; 0 = off)
    std::atomic<int> a4_group_m{0};     // "a4_group_m": row tiles per XCD group of the 4-wave kernel; 0 = by K
    std::atomic<int> conv_nb{0};        // "conv_nb": N-tile width of the implicit-GEMM convolution; 0 = by Cout, 1 / 2 forced
    std::atomic<int> conv_gather{0};    // "conv_gather": 1 = the general (pointer-free) gather
    std::atomic<int> vae_rms3{1};       // "vae_rms3": 0 = RMS_norm+SiLU at C = 96 / 192 / 384 on the power-of-two kernel
    std::atomic<int> conv_direct{1};    // "conv_direct": 0 = the 96-channel 3x3x3 / 192-channel upsample convolutions on the implicit GEMM
};
// gf_conv_direct.hip: direct convolution of the 96-channel level; GF_ERR_UNSUPPORTED = shape not covered (caller falls back)
int gf_conv3d_direct_c96(const void* src_walk, const void* Wm, int64_t ldw, const void* bias, void* out, int64_t T_out, int64_t H,
                         int64_t W, int64_t N, int epilogue, const void* resid, const void* zero_page, void* stream);
int gf_conv2d_up_direct_c192(const void* src0, const void* Wm, int64_t ldw, const void* bias, void* out, int64_t T_out, int64_t Hs,
                             int64_t Ws, const void* zero_page, void* stream);
const GfOptions& gf_options();          // gf_abi.hip

// One-time, per-DEVICE, thread-safe launcher setup (hipFuncSetAttribute is a property of the function on the current
// device: a process that drives a second GPU must set it there too).  `f()` runs at most once per device and returns
// a hipError_t; a failure is not latched (the next call retries).
#include <atomic>
#include <mutex>
struct GfDeviceOnce {
    std::atomic<uint64_t> done{0};
    std::mutex mu;
};
template <class F>
static inline hipError_t gf_once_per_device(GfDeviceOnce& st, F&& f) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev < 0 || dev >= 64) return hipErrorInvalidDevice;
    const uint64_t bit = 1ull << dev;
    if (st.done.load(std::memory_order_acquire) & bit) return hipSuccess;
    std::lock_guard<std::mutex> lock(st.mu);
    if (st.done.load(std::memory_order_relaxed) & bit) return hipSuccess;
    e = f();
    if (e == hipSuccess) st.done.fetch_or(bit, std::memory_order_release);
    return e;
}
