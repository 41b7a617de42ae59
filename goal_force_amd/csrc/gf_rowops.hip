// gf_rowops.hip — HBM-bound row kernels of the DiT block: LayerNorm(+modulate) and
// full-width RMSNorm(+3-D RoPE).  One workgroup of 128 threads per token row; the row
// (<= 8192 bf16) lives in registers between the reduction and the write, so every byte is
// read once and written once (algorithmic bytes: 2 * rows * dim * 2 B).
//
// Reference semantics (file:line relative to the reference tree):
//   LayerNorm fp32 math + one rounding ......... diffsynth/vram_management/layers.py:78-92
//   modulate x*(1+scale)+shift (bf16 eager) ..... diffsynth/models/wan_video_dit.py:64-65
//   RMSNorm .float() norm, .to(bf16), * weight .. diffsynth/models/wan_video_dit.py:100-111
//   rope_apply on adjacent pairs ................ diffsynth/models/wan_video_dit.py:92-97
#include "gf_common.h"

namespace {

constexpr int ROW_THREADS = 128;
constexpr int ROW_NCH = 8;  // 16-byte chunks per thread -> dim <= 128*8*8 = 8192

__global__ __launch_bounds__(ROW_THREADS) void layernorm_modulate_kernel(
    const u16* __restrict__ x, u16* __restrict__ out, const u16* __restrict__ weight,
    const u16* __restrict__ bias, const u16* __restrict__ scale1p, const u16* __restrict__ shift,
    int dim, long x_stride, long out_stride, float eps) {
    __shared__ float red[ROW_THREADS / 64];
    const long row = blockIdx.x;
    const int tid = threadIdx.x;
    const int nchunks = dim >> 3;
    const u16* xr = x + row * x_stride;
    u16x8 v[ROW_NCH];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < ROW_NCH; ++i) {
        const int c = tid + i * ROW_THREADS;
        if (c < nchunks) {
            v[i] = *reinterpret_cast<const u16x8*>(xr + (c << 3));
#pragma unroll
            for (int j = 0; j < 8; ++j) s += bf2f(v[i][j]);
        }
    }
    const float mean = block_sum<ROW_THREADS>(s, red) / (float)dim;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < ROW_NCH; ++i) {
        const int c = tid + i * ROW_THREADS;
        if (c < nchunks) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float d = bf2f(v[i][j]) - mean;
                q += d * d;
            }
        }
    }
    const float var = block_sum<ROW_THREADS>(q, red) / (float)dim;
    const float rstd = 1.0f / sqrtf(var + eps);
    u16* orow = out + row * out_stride;
#pragma unroll
    for (int i = 0; i < ROW_NCH; ++i) {
        const int c = tid + i * ROW_THREADS;
        if (c < nchunks) {
            u16x8 w8, b8, sc8, sh8;
            if (weight) w8 = *reinterpret_cast<const u16x8*>(weight + (c << 3));
            if (bias) b8 = *reinterpret_cast<const u16x8*>(bias + (c << 3));
            if (scale1p) sc8 = *reinterpret_cast<const u16x8*>(scale1p + (c << 3));
            if (shift) sh8 = *reinterpret_cast<const u16x8*>(shift + (c << 3));
            u16x8 o;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float y = (bf2f(v[i][j]) - mean) * rstd;
                if (weight) y = y * bf2f(w8[j]);
                if (bias) y = y + bf2f(b8[j]);
                y = rbf(y);                                   // .type_as(x)
                if (scale1p) y = rbf(y * bf2f(sc8[j]));       // x * (1 + scale)
                if (shift) y = rbf(y + bf2f(sh8[j]));         // + shift
                o[j] = f2bf(y);
            }
            *reinterpret_cast<u16x8*>(orow + (c << 3)) = o;
        }
    }
}

__global__ __launch_bounds__(ROW_THREADS) void rmsnorm_rope_kernel(
    u16* __restrict__ x, const u16* __restrict__ weight, const float* __restrict__ cos_tab,
    const float* __restrict__ sin_tab, int dim, int head_dim, long x_stride, float eps) {
    __shared__ float red[ROW_THREADS / 64];
    const long row = blockIdx.x;
    const int tid = threadIdx.x;
    const int nchunks = dim >> 3;
    u16* xr = x + row * x_stride;
    u16x8 v[ROW_NCH];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < ROW_NCH; ++i) {
        const int c = tid + i * ROW_THREADS;
        if (c < nchunks) {
            v[i] = *reinterpret_cast<const u16x8*>(xr + (c << 3));
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float f = bf2f(v[i][j]);
                s += f * f;
            }
        }
    }
    const float ms = block_sum<ROW_THREADS>(s, red) / (float)dim;
    const float rstd = 1.0f / sqrtf(ms + eps);
    const int half = head_dim >> 1;
#pragma unroll
    for (int i = 0; i < ROW_NCH; ++i) {
        const int c = tid + i * ROW_THREADS;
        if (c < nchunks) {
            const u16x8 w8 = *reinterpret_cast<const u16x8*>(weight + (c << 3));
            float y[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float n = rbf(bf2f(v[i][j]) * rstd);  // norm(x.float()).to(dtype)
                y[j] = rbf(n * bf2f(w8[j]));                // * weight (bf16 multiply)
            }
            u16x8 o;
            if (cos_tab) {
                const int p0 = ((c << 3) % head_dim) >> 1;  // first complex pair of this chunk
                const f32x4 cs = *reinterpret_cast<const f32x4*>(cos_tab + row * half + p0);
                const f32x4 sn = *reinterpret_cast<const f32x4*>(sin_tab + row * half + p0);
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    const float a = y[2 * p], b = y[2 * p + 1];
                    o[2 * p] = f2bf(__builtin_fmaf(a, cs[p], -(b * sn[p])));      // the wave kernel's explicit contraction
                    o[2 * p + 1] = f2bf(__builtin_fmaf(a, sn[p], b * cs[p]));
                }
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) o[j] = f2bf(y[j]);
            }
            *reinterpret_cast<u16x8*>(xr + (c << 3)) = o;
        }
    }
}

// ---- one WAVE per row (dim = NCH * 512): the row lives in one wave's registers, both reductions are wave shuffles, no LDS,
// no barrier, no ragged-chunk predicates; 4 rows per 256-thread workgroup.  Used for the DiT widths (5120, 4096, 1536);
// the 128-thread kernels above stay for every other width.
constexpr int WROWS = 4;

template <int NCH>
__global__ __launch_bounds__(64 * WROWS) void layernorm_modulate_wave_kernel(
    const u16* __restrict__ x, u16* __restrict__ out, const u16* __restrict__ weight, const u16* __restrict__ bias,
    const u16* __restrict__ scale1p, const u16* __restrict__ shift, long rows, long x_stride, long out_stride, float eps) {
    constexpr int DIM = NCH * 512;
    const long row = (long)blockIdx.x * WROWS + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int lane = threadIdx.x & 63;
    const u16* xr = x + row * x_stride + lane * 8;
    u16x8 v[NCH];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        v[i] = *reinterpret_cast<const u16x8*>(xr + i * 512);
#pragma unroll
        for (int j = 0; j < 8; ++j) s += bf2f(v[i][j]);
    }
    const float mean = wave_sum(s) * (1.0f / DIM);
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float d = bf2f(v[i][j]) - mean;
            q += d * d;
        }
    const float rstd = 1.0f / sqrtf(wave_sum(q) * (1.0f / DIM) + eps);
    u16* orow = out + row * out_stride + lane * 8;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int c0 = i * 512 + lane * 8;
        u16x8 w8, b8, sc8, sh8, o;
        if (weight) w8 = *reinterpret_cast<const u16x8*>(weight + c0);
        if (bias) b8 = *reinterpret_cast<const u16x8*>(bias + c0);
        if (scale1p) sc8 = *reinterpret_cast<const u16x8*>(scale1p + c0);
        if (shift) sh8 = *reinterpret_cast<const u16x8*>(shift + c0);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float y = (bf2f(v[i][j]) - mean) * rstd;
            if (weight) y = y * bf2f(w8[j]);
            if (bias) y = y + bf2f(b8[j]);
            y = rbf(y);                                   // .type_as(x)
            if (scale1p) y = rbf(y * bf2f(sc8[j]));       // x * (1 + scale)
            if (shift) y = rbf(y + bf2f(sh8[j]));         // + shift
            o[j] = f2bf(y);
        }
        *reinterpret_cast<u16x8*>(orow + i * 512) = o;
    }
}

// ---- the same wave-per-row LayerNorm, specialised (round 3).  The kernel above decides per ELEMENT which of weight / bias / scale /
// shift exist (166 v_cndmask per row and lane) and rounds every value on its own (one v_cvt_pk_bf16_f32 + a shift per value and
// rounding): 1847 vector instructions per row and lane = 23 per element — enough to hold an HBM-bound kernel at 5.0 TB/s.  Here
// the operand set is a template parameter and values go through the roundings in PAIRS (one v_cvt_pk_bf16_f32 per pair, unpacked
// by a shift and a mask; multiplies / adds on float2 so that the packed-math forms can be used): the same operations in the same
// order on every value, so the same bits, at about half the instructions.
//   MODE 0: LayerNorm                      (norm without affine, no modulation)
//   MODE 1: LayerNorm * a + b              (a = weight, b = bias: norm3)
//   MODE 2: modulate(LayerNorm, b, a)      (a = 1 + scale, b = shift: norm1 / norm2 / head, the reference's three bf16 roundings)
//   FP8: the result is quantised in registers for an fp8_linear consumer (layernorm_modulate_fp8_wave_kernel's contract).
__device__ __forceinline__ gf_f32x2 unpack2bf(unsigned u) { return gf_f32x2{__uint_as_float(u << 16), __uint_as_float(u & 0xffff0000u)}; }
__device__ __forceinline__ gf_f32x2 ln2_fma(gf_f32x2 a, gf_f32x2 b, gf_f32x2 c) {
    return __builtin_elementwise_fma(a, b, c);
}
// (contract off: a product must stay a product — the roundings of the reference's separate ops are the point)
__device__ __forceinline__ gf_f32x2 ln2_mul(gf_f32x2 a, gf_f32x2 b) {
#pragma clang fp contract(off)
    return a * b;
}
__device__ __forceinline__ gf_f32x2 ln2_add(gf_f32x2 a, gf_f32x2 b) {
#pragma clang fp contract(off)
    return a + b;
}

template <int NCH, int MODE, bool FP8>
__global__ __launch_bounds__(64 * WROWS) void layernorm_wave2_kernel(const u16* __restrict__ x, void* __restrict__ outp,
                                                                     float* __restrict__ scale_out, const u16* __restrict__ avec,
                                                                     const u16* __restrict__ bvec, long rows, long x_stride,
                                                                     long out_stride, float eps) {
    constexpr int DIM = NCH * 512;
    const long row = (long)blockIdx.x * WROWS + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int lane = threadIdx.x & 63;
    const u16* xr = x + row * x_stride + lane * 8;
    u32x4 v[NCH];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        v[i] = *reinterpret_cast<const u32x4*>(xr + i * 512);
#pragma unroll
        for (int p = 0; p < 4; ++p) {           // element order 0..7 of the chunk, as the per-element kernel adds them
            const gf_f32x2 f = unpack2bf(v[i][p]);
            s += f[0];
            s += f[1];
        }
    }
    // x - mean as ONE fma on the row total, fma(total, -1/DIM, x), and the variance as fma(sum, 1/DIM, eps): what the per-element
    // kernel's expressions contract to (hipcc's default fp-contract), written out so that both kernels round alike
    const float tot = wave_sum(s);
    const gf_f32x2 tot2 = {tot, tot}, ninv2 = {-1.0f / DIM, -1.0f / DIM};
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i)
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const gf_f32x2 d = ln2_fma(tot2, ninv2, unpack2bf(v[i][p]));
            q = __builtin_fmaf(d[0], d[0], q);
            q = __builtin_fmaf(d[1], d[1], q);
        }
    const float rstd = 1.0f / sqrtf(__builtin_fmaf(wave_sum(q), 1.0f / DIM, eps));
    const gf_f32x2 rstd2 = {rstd, rstd};
    __attribute__((ext_vector_type(2))) unsigned short mxb = {0, 0};
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int c0 = i * 512 + lane * 8;
        u32x4 a8 = {0, 0, 0, 0}, b8 = {0, 0, 0, 0};
        if constexpr (MODE != 0) {
            a8 = *reinterpret_cast<const u32x4*>(avec + c0);
            b8 = *reinterpret_cast<const u32x4*>(bvec + c0);
        }
        u32x4 o;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            gf_f32x2 y = ln2_mul(ln2_fma(tot2, ninv2, unpack2bf(v[i][p])), rstd2);
            unsigned r;
            if constexpr (MODE == 1) {
                y = ln2_add(ln2_mul(y, unpack2bf(a8[p])), unpack2bf(b8[p]));   // * weight, + bias: two roundings, as the per-element kernel
                r = pack2bf(y[0], y[1]);                                         // (its selects keep the compiler from fusing them); .type_as(x)
            } else if constexpr (MODE == 2) {
                r = pack2bf(y[0], y[1]);                                                 // .type_as(x)
                y = ln2_mul(unpack2bf(r), unpack2bf(a8[p]));                             // x * (1 + scale)
                r = pack2bf(y[0], y[1]);
                y = ln2_add(unpack2bf(r), unpack2bf(b8[p]));                             // + shift
                r = pack2bf(y[0], y[1]);
            } else {
                r = pack2bf(y[0], y[1]);
            }
            o[p] = r;
            if constexpr (FP8) {   // |bf16| orders like its 15-bit pattern: the row maximum on the packed pairs (one and + one v_pk_max_u16)
                typedef __attribute__((ext_vector_type(2))) unsigned short us2;
                mxb = __builtin_elementwise_max(mxb, __builtin_bit_cast(us2, r & 0x7fff7fffu));
            }
        }
        if constexpr (FP8) v[i] = o;            // the bf16 row fp8_linear receives, kept in registers
        else *reinterpret_cast<u32x4*>((u16*)outp + row * out_stride + lane * 8 + i * 512) = o;
    }
    if constexpr (FP8) {
        float mx = bf2f(mxb[0] > mxb[1] ? mxb[0] : mxb[1]);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
        const float sc = fmaxf(rbf(mx / 448.0f), 1.0f);      // x_max is bf16 in the reference: x_max / 448 rounds to bf16 before the clamp
        if (lane == 0) scale_out[row] = sc;
        const float den = sc + 1e-8f;
        const bool unit = den == 1.0f;                       // wave-uniform; 1 + 1e-8 IS 1 in fp32, so x / den = x exactly
        unsigned char* orow = (unsigned char*)outp + row * out_stride + lane * 8;
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            unsigned w[2];
#pragma unroll
            for (int h2 = 0; h2 < 2; ++h2) {
                gf_f32x2 f0 = unpack2bf(v[i][2 * h2]), f1 = unpack2bf(v[i][2 * h2 + 1]);
                if (!unit) {
                    f0 = gf_f32x2{f0[0] / den, f0[1] / den};
                    f1 = gf_f32x2{f1[0] / den, f1[1] / den};
                }
                unsigned t = 0;
                t = __builtin_amdgcn_cvt_pk_fp8_f32(f0[0], f0[1], t, false);
                t = __builtin_amdgcn_cvt_pk_fp8_f32(f1[0], f1[1], t, true);
                w[h2] = t;
            }
            *reinterpret_cast<u32x2*>(orow + i * 512) = u32x2{w[0], w[1]};
        }
    }
}

// ROPE: 0 = no rotation, 1 = head_dim divides 512: a lane's four complex pairs are the same in every 512-element chunk
// (c0 % head_dim = 8 lane % head_dim), so its cos / sin are loaded ONCE per row instead of once per chunk — 18 loads and ten
// integer modulo sequences less per row and wave; 2 = any head_dim (per chunk).
template <int NCH, int ROPE>
__global__ __launch_bounds__(64 * WROWS) void rmsnorm_rope_wave_kernel(u16* __restrict__ x, const u16* __restrict__ weight,
                                                                         const float* __restrict__ cos_tab,
                                                                         const float* __restrict__ sin_tab, long rows,
                                                                         int head_dim, long x_stride, float eps) {
    constexpr int DIM = NCH * 512;
    const long row = (long)blockIdx.x * WROWS + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int lane = threadIdx.x & 63;
    u16* xr = x + row * x_stride + lane * 8;
    u16x8 v[NCH];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        v[i] = *reinterpret_cast<const u16x8*>(xr + i * 512);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float f = bf2f(v[i][j]);
            s += f * f;
        }
    }
    const float rstd = 1.0f / sqrtf(wave_sum(s) * (1.0f / DIM) + eps);
    const int half = head_dim >> 1;
    f32x4 cs1 = {0.f, 0.f, 0.f, 0.f}, sn1 = {0.f, 0.f, 0.f, 0.f};
    if constexpr (ROPE == 1) {
        const int p0 = ((lane * 8) % head_dim) >> 1;
        cs1 = *reinterpret_cast<const f32x4*>(cos_tab + row * half + p0);
        sn1 = *reinterpret_cast<const f32x4*>(sin_tab + row * half + p0);
    }
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int c0 = i * 512 + lane * 8;
        const u16x8 w8 = *reinterpret_cast<const u16x8*>(weight + c0);
        float y[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float n = rbf(bf2f(v[i][j]) * rstd);  // norm(x.float()).to(dtype)
            y[j] = rbf(n * bf2f(w8[j]));                // * weight (bf16 multiply)
        }
        u16x8 o;
        if constexpr (ROPE != 0) {
            f32x4 cs = cs1, sn = sn1;
            if constexpr (ROPE == 2) {
                const int p0 = (c0 % head_dim) >> 1;        // first complex pair of this chunk
                cs = *reinterpret_cast<const f32x4*>(cos_tab + row * half + p0);
                sn = *reinterpret_cast<const f32x4*>(sin_tab + row * half + p0);
            }
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                // (a + i b)(cos + i sin); the contraction is written out so that every build and both ROPE modes round alike
                const float a = y[2 * p], b = y[2 * p + 1];
                o[2 * p] = f2bf(__builtin_fmaf(a, cs[p], -(b * sn[p])));
                o[2 * p + 1] = f2bf(__builtin_fmaf(a, sn[p], b * cs[p]));
            }
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = f2bf(y[j]);
        }
        *reinterpret_cast<u16x8*>(xr + i * 512) = o;
    }
}

// fp8 activation quantisation of AutoWrappedLinear.fp8_linear (VRAM:115-140):
//   scale_a = clamp(rowmax|x| / 448, min=1)  (fp32);   x8 = e4m3(float(x) / (scale_a + 1e-8))
// OCP e4m3fn on gfx950 (not MI300's fnuz).  One workgroup per row, row kept in registers.
__global__ __launch_bounds__(ROW_THREADS) void quant_fp8_rowscale_kernel(const u16* __restrict__ x,
                                                                         unsigned char* __restrict__ out,
                                                                         float* __restrict__ scale, int dim,
                                                                         long x_stride, long out_stride) {
    __shared__ float red[ROW_THREADS / 64];
    const long row = blockIdx.x;
    const int tid = threadIdx.x;
    const int nchunks = dim >> 3;
    const u16* xr = x + row * x_stride;
    constexpr int NCH = 14;  // dim <= 128*14*8 = 14336 (FFN hidden 13824)
    u16x8 v[NCH];
    float mx = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int c = tid + i * ROW_THREADS;
        if (c < nchunks) {
            v[i] = *reinterpret_cast<const u16x8*>(xr + (c << 3));
#pragma unroll
            for (int j = 0; j < 8; ++j) mx = fmaxf(mx, fabsf(bf2f(v[i][j])));
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    if ((tid & 63) == 0) red[tid >> 6] = mx;
    __syncthreads();
    mx = fmaxf(red[0], red[1]);
    // x_max is a bf16 tensor in the reference, so x_max / 448 is rounded to bf16 before clamp(min=1).float()
    const float sc = fmaxf(rbf(mx / 448.0f), 1.0f);
    if (tid == 0) scale[row] = sc;
    const float den = sc + 1e-8f;
    unsigned char* orow = out + row * out_stride;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int c = tid + i * ROW_THREADS;
        if (c < nchunks) {
            float f[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) f[j] = bf2f(v[i][j]) / den;
            unsigned w0 = 0, w1 = 0;
            w0 = __builtin_amdgcn_cvt_pk_fp8_f32(f[0], f[1], w0, false);
            w0 = __builtin_amdgcn_cvt_pk_fp8_f32(f[2], f[3], w0, true);
            w1 = __builtin_amdgcn_cvt_pk_fp8_f32(f[4], f[5], w1, false);
            w1 = __builtin_amdgcn_cvt_pk_fp8_f32(f[6], f[7], w1, true);
            *reinterpret_cast<u32x2*>(orow + (c << 3)) = u32x2{w0, w1};
        }
    }
}

// LayerNorm(+affine)(+modulate) whose consumer is an fp8 Linear (BASELINE config 5): the bf16 row the reference would hand to
// fp8_linear never goes to HBM — the wave that normalised it takes its row maximum (wave shuffles), forms scale_a and writes the
// e4m3 bytes + the scale.  Same arithmetic, in the same order, as layernorm_modulate_wave_kernel followed by
// quant_fp8_rowscale_kernel (bit-identical: tests/test_fp8.py), one 335 MB write and one 335 MB read less per use.
template <int NCH>
__global__ __launch_bounds__(64 * WROWS) void layernorm_modulate_fp8_wave_kernel(
    const u16* __restrict__ x, unsigned char* __restrict__ out8, float* __restrict__ scale, const u16* __restrict__ weight,
    const u16* __restrict__ bias, const u16* __restrict__ scale1p, const u16* __restrict__ shift, long rows, long x_stride,
    long out_stride, float eps) {
    constexpr int DIM = NCH * 512;
    const long row = (long)blockIdx.x * WROWS + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int lane = threadIdx.x & 63;
    const u16* xr = x + row * x_stride + lane * 8;
    u16x8 v[NCH];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        v[i] = *reinterpret_cast<const u16x8*>(xr + i * 512);
#pragma unroll
        for (int j = 0; j < 8; ++j) s += bf2f(v[i][j]);
    }
    const float mean = wave_sum(s) * (1.0f / DIM);
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float d = bf2f(v[i][j]) - mean;
            q += d * d;
        }
    const float rstd = 1.0f / sqrtf(wave_sum(q) * (1.0f / DIM) + eps);
    float mx = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int c0 = i * 512 + lane * 8;
        u16x8 w8, b8, sc8, sh8;
        if (weight) w8 = *reinterpret_cast<const u16x8*>(weight + c0);
        if (bias) b8 = *reinterpret_cast<const u16x8*>(bias + c0);
        if (scale1p) sc8 = *reinterpret_cast<const u16x8*>(scale1p + c0);
        if (shift) sh8 = *reinterpret_cast<const u16x8*>(shift + c0);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float y = (bf2f(v[i][j]) - mean) * rstd;
            if (weight) y = y * bf2f(w8[j]);
            if (bias) y = y + bf2f(b8[j]);
            y = rbf(y);                                   // .type_as(x)
            if (scale1p) y = rbf(y * bf2f(sc8[j]));       // x * (1 + scale)
            if (shift) y = rbf(y + bf2f(sh8[j]));         // + shift
            v[i][j] = f2bf(y);                            // the bf16 row fp8_linear receives (kept in registers)
            mx = fmaxf(mx, fabsf(y));
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    const float sc = fmaxf(rbf(mx / 448.0f), 1.0f);      // x_max is bf16 in the reference: x_max / 448 rounds to bf16 before the clamp
    if (lane == 0) scale[row] = sc;
    const float den = sc + 1e-8f;
    const bool unit = den == 1.0f;                       // wave-uniform; 1 + 1e-8 IS 1 in fp32, so x / den = x exactly
    unsigned char* orow = out8 + row * out_stride + lane * 8;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        float f[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) f[j] = unit ? bf2f(v[i][j]) : bf2f(v[i][j]) / den;
        unsigned w0 = 0, w1 = 0;
        w0 = __builtin_amdgcn_cvt_pk_fp8_f32(f[0], f[1], w0, false);
        w0 = __builtin_amdgcn_cvt_pk_fp8_f32(f[2], f[3], w0, true);
        w1 = __builtin_amdgcn_cvt_pk_fp8_f32(f[4], f[5], w1, false);
        w1 = __builtin_amdgcn_cvt_pk_fp8_f32(f[6], f[7], w1, true);
        *reinterpret_cast<u32x2*>(orow + i * 512) = u32x2{w0, w1};
    }
}

// plain bf16 -> e4m3 cast (the weight side of fp8_linear: `weight.to(float8_e4m3fn)`, unit scale, VRAM:138)
__global__ __launch_bounds__(256) void cast_fp8_kernel(const u16* __restrict__ x, unsigned char* __restrict__ out, long n8) {
    const long stride = (long)gridDim.x * 256;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n8; i += stride) {
        const u16x8 v = reinterpret_cast<const u16x8*>(x)[i];
        unsigned w0 = 0, w1 = 0;
        w0 = __builtin_amdgcn_cvt_pk_fp8_f32(bf2f(v[0]), bf2f(v[1]), w0, false);
        w0 = __builtin_amdgcn_cvt_pk_fp8_f32(bf2f(v[2]), bf2f(v[3]), w0, true);
        w1 = __builtin_amdgcn_cvt_pk_fp8_f32(bf2f(v[4]), bf2f(v[5]), w1, false);
        w1 = __builtin_amdgcn_cvt_pk_fp8_f32(bf2f(v[6]), bf2f(v[7]), w1, true);
        reinterpret_cast<u32x2*>(out)[i] = u32x2{w0, w1};
    }
}

}  // namespace

extern "C" GF_API int gf_quant_fp8_rowscale(const void* x, void* out8, float* scale, int64_t rows, int64_t dim,
                                            int64_t x_stride, int64_t out_stride, void* stream) {
    GF_CHECK_ARG(x && out8 && scale && rows >= 0, "gf_quant_fp8_rowscale: null pointer");
    GF_CHECK_ARG(dim > 0 && dim % 8 == 0 && dim <= ROW_THREADS * 14 * 8, "gf_quant_fp8_rowscale: dim=%ld must be a multiple of 8 and <= %d",
                 (long)dim, ROW_THREADS * 14 * 8);
    GF_CHECK_ARG(x_stride % 8 == 0 && out_stride % 8 == 0 && gf_aligned16(x) && (((uintptr_t)out8) & 7u) == 0,
                 "gf_quant_fp8_rowscale: alignment");
    if (rows == 0) return GF_OK;
    hipLaunchKernelGGL(quant_fp8_rowscale_kernel, dim3((unsigned)rows), dim3(ROW_THREADS), 0, (hipStream_t)stream,
                       (const u16*)x, (unsigned char*)out8, scale, (int)dim, (long)x_stride, (long)out_stride);
    GF_CHECK_LAUNCH("gf_quant_fp8_rowscale");
    return GF_OK;
}

extern "C" GF_API int gf_cast_fp8(const void* x, void* out8, int64_t n, void* stream) {
    GF_CHECK_ARG(x && out8 && n >= 0 && n % 8 == 0, "gf_cast_fp8: n must be a multiple of 8");
    GF_CHECK_ARG(gf_aligned16(x) && (((uintptr_t)out8) & 7u) == 0, "gf_cast_fp8: alignment");
    if (n == 0) return GF_OK;
    long blocks = (n / 8 + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(cast_fp8_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const u16*)x,
                       (unsigned char*)out8, (long)(n / 8));
    GF_CHECK_LAUNCH("gf_cast_fp8");
    return GF_OK;
}

extern "C" GF_API int gf_layernorm_modulate(const void* x, void* out, const void* weight, const void* bias,
                                     const void* scale1p, const void* shift, int64_t rows, int64_t dim,
                                     int64_t x_stride, int64_t out_stride, float eps, void* stream) {
    GF_CHECK_ARG(x && out, "gf_layernorm_modulate: null x/out");
    GF_CHECK_ARG(rows >= 0 && dim > 0 && dim % 8 == 0 && dim <= ROW_THREADS * ROW_NCH * 8,
                 "gf_layernorm_modulate: dim=%ld must be a multiple of 8 and <= %d", (long)dim,
                 ROW_THREADS * ROW_NCH * 8);
    GF_CHECK_ARG(x_stride % 8 == 0 && out_stride % 8 == 0 && gf_aligned16(x) && gf_aligned16(out),
                 "gf_layernorm_modulate: rows must be 16-byte aligned");
    GF_CHECK_ARG((!weight || gf_aligned16(weight)) && (!bias || gf_aligned16(bias)) &&
                     (!scale1p || gf_aligned16(scale1p)) && (!shift || gf_aligned16(shift)),
                 "gf_layernorm_modulate: vectors must be 16-byte aligned");
    if (rows == 0) return GF_OK;
    {   // the three operand sets of the DiT (plain, weight + bias, scale + shift) at the wave-per-row widths: the specialised kernel
        const int mode = (!weight && !bias && !scale1p && !shift) ? 0 : (weight && bias && !scale1p && !shift) ? 1
                         : (!weight && !bias && scale1p && shift) ? 2 : -1;
        const void* av = mode == 1 ? weight : scale1p;
        const void* bv = mode == 1 ? bias : shift;
        const dim3 grid_((unsigned)((rows + WROWS - 1) / WROWS)), block_(64 * WROWS);
#define GF_LN2(NCH, M)                                                                                                   \
    if (dim == NCH * 512 && mode == M) {                                                                                 \
        hipLaunchKernelGGL((layernorm_wave2_kernel<NCH, M, false>), grid_, block_, 0, (hipStream_t)stream, (const u16*)x, out,     \
                           (float*)nullptr, (const u16*)av, (const u16*)bv, (long)rows, (long)x_stride, (long)out_stride, eps);     \
        GF_CHECK_LAUNCH("gf_layernorm_modulate");                                                                        \
        return GF_OK;                                                                                                    \
    }
        GF_LN2(10, 0) GF_LN2(10, 1) GF_LN2(10, 2) GF_LN2(8, 0) GF_LN2(8, 1) GF_LN2(8, 2) GF_LN2(3, 0) GF_LN2(3, 1) GF_LN2(3, 2)
#undef GF_LN2
    }
#define GF_LN_WAVE(NCH)                                                                                                  \
    if (dim == NCH * 512) {                                                                                              \
        hipLaunchKernelGGL(layernorm_modulate_wave_kernel<NCH>, dim3((unsigned)((rows + WROWS - 1) / WROWS)),            \
                           dim3(64 * WROWS), 0, (hipStream_t)stream, (const u16*)x, (u16*)out, (const u16*)weight,       \
                           (const u16*)bias, (const u16*)scale1p, (const u16*)shift, (long)rows, (long)x_stride,        \
                           (long)out_stride, eps);                                                                       \
        GF_CHECK_LAUNCH("gf_layernorm_modulate");                                                                        \
        return GF_OK;                                                                                                    \
    }
    GF_LN_WAVE(10) GF_LN_WAVE(8) GF_LN_WAVE(3)
#undef GF_LN_WAVE
    hipLaunchKernelGGL(layernorm_modulate_kernel, dim3((unsigned)rows), dim3(ROW_THREADS), 0,
                       (hipStream_t)stream, (const u16*)x, (u16*)out, (const u16*)weight, (const u16*)bias,
                       (const u16*)scale1p, (const u16*)shift, (int)dim, (long)x_stride, (long)out_stride, eps);
    GF_CHECK_LAUNCH("gf_layernorm_modulate");
    return GF_OK;
}

extern "C" GF_API int gf_layernorm_modulate_fp8(const void* x, void* out8, float* scale, const void* weight, const void* bias,
                                                const void* scale1p, const void* shift, int64_t rows, int64_t dim,
                                                int64_t x_stride, int64_t out_stride, float eps, void* stream) {
    GF_CHECK_ARG(x && out8 && scale, "gf_layernorm_modulate_fp8: null x/out8/scale");
    GF_CHECK_ARG(rows >= 0 && (dim == 5120 || dim == 4096 || dim == 1536),
                 "gf_layernorm_modulate_fp8: dim=%ld is not one of the wave-per-row widths (5120, 4096, 1536); use "
                 "gf_layernorm_modulate + gf_quant_fp8_rowscale", (long)dim);
    GF_CHECK_ARG(x_stride % 8 == 0 && out_stride % 8 == 0 && gf_aligned16(x) && (((uintptr_t)out8) & 7u) == 0,
                 "gf_layernorm_modulate_fp8: rows must be 16-byte (x) / 8-byte (out8) aligned");
    GF_CHECK_ARG((!weight || gf_aligned16(weight)) && (!bias || gf_aligned16(bias)) &&
                     (!scale1p || gf_aligned16(scale1p)) && (!shift || gf_aligned16(shift)),
                 "gf_layernorm_modulate_fp8: vectors must be 16-byte aligned");
    if (rows == 0) return GF_OK;
    {
        const int mode = (!weight && !bias && !scale1p && !shift) ? 0 : (weight && bias && !scale1p && !shift) ? 1
                         : (!weight && !bias && scale1p && shift) ? 2 : -1;
        const void* av = mode == 1 ? weight : scale1p;
        const void* bv = mode == 1 ? bias : shift;
        const dim3 grid_((unsigned)((rows + WROWS - 1) / WROWS)), block_(64 * WROWS);
#define GF_LN2(NCH, M)                                                                                                   \
    if (dim == NCH * 512 && mode == M) {                                                                                 \
        hipLaunchKernelGGL((layernorm_wave2_kernel<NCH, M, true>), grid_, block_, 0, (hipStream_t)stream, (const u16*)x, out8,     \
                           scale, (const u16*)av, (const u16*)bv, (long)rows, (long)x_stride, (long)out_stride, eps);   \
        GF_CHECK_LAUNCH("gf_layernorm_modulate_fp8");                                                                    \
        return GF_OK;                                                                                                    \
    }
        GF_LN2(10, 0) GF_LN2(10, 1) GF_LN2(10, 2) GF_LN2(8, 0) GF_LN2(8, 1) GF_LN2(8, 2) GF_LN2(3, 0) GF_LN2(3, 1) GF_LN2(3, 2)
#undef GF_LN2
    }
#define GF_LN8_WAVE(NCH)                                                                                                 \
    if (dim == NCH * 512) {                                                                                              \
        hipLaunchKernelGGL(layernorm_modulate_fp8_wave_kernel<NCH>, dim3((unsigned)((rows + WROWS - 1) / WROWS)),        \
                           dim3(64 * WROWS), 0, (hipStream_t)stream, (const u16*)x, (unsigned char*)out8, scale,          \
                           (const u16*)weight, (const u16*)bias, (const u16*)scale1p, (const u16*)shift, (long)rows,     \
                           (long)x_stride, (long)out_stride, eps);                                                       \
    }
    GF_LN8_WAVE(10) GF_LN8_WAVE(8) GF_LN8_WAVE(3)
#undef GF_LN8_WAVE
    GF_CHECK_LAUNCH("gf_layernorm_modulate_fp8");
    return GF_OK;
}

extern "C" GF_API int gf_rmsnorm_rope(void* x, const void* weight, const float* cos_tab, const float* sin_tab,
                               int64_t rows, int64_t dim, int64_t head_dim, int64_t x_stride, float eps,
                               void* stream) {
    GF_CHECK_ARG(x && weight, "gf_rmsnorm_rope: null x/weight");
    GF_CHECK_ARG(rows >= 0 && dim > 0 && dim % 8 == 0 && dim <= ROW_THREADS * ROW_NCH * 8,
                 "gf_rmsnorm_rope: dim=%ld must be a multiple of 8 and <= %d", (long)dim,
                 ROW_THREADS * ROW_NCH * 8);
    GF_CHECK_ARG((cos_tab == nullptr) == (sin_tab == nullptr), "gf_rmsnorm_rope: cos/sin must both be set or both NULL");
    GF_CHECK_ARG(head_dim > 0 && head_dim % 8 == 0 && dim % head_dim == 0,
                 "gf_rmsnorm_rope: head_dim=%ld must be a multiple of 8 dividing dim", (long)head_dim);
    GF_CHECK_ARG(x_stride % 8 == 0 && gf_aligned16(x) && gf_aligned16(weight) &&
                     (!cos_tab || (gf_aligned16(cos_tab) && gf_aligned16(sin_tab))),
                 "gf_rmsnorm_rope: 16-byte alignment required");
    if (rows == 0) return GF_OK;
#define GF_RMS_WAVE_ROPE(NCH, ROPE)                                                                                      \
    if (rope_ == ROPE)                                                                                                   \
        hipLaunchKernelGGL((rmsnorm_rope_wave_kernel<NCH, ROPE>), grid_, block_, 0, (hipStream_t)stream, (u16*)x,        \
                           (const u16*)weight, cos_tab, sin_tab, (long)rows, (int)head_dim, (long)x_stride, eps);
#define GF_RMS_WAVE(NCH)                                                                                                 \
    if (dim == NCH * 512) {                                                                                              \
        const dim3 grid_((unsigned)((rows + WROWS - 1) / WROWS)), block_(64 * WROWS);                                     \
        const int rope_ = !cos_tab ? 0 : (512 % head_dim == 0 ? 1 : 2);                                                  \
        GF_RMS_WAVE_ROPE(NCH, 0) GF_RMS_WAVE_ROPE(NCH, 1) GF_RMS_WAVE_ROPE(NCH, 2)                                        \
        GF_CHECK_LAUNCH("gf_rmsnorm_rope");                                                                              \
        return GF_OK;                                                                                                    \
    }
    GF_RMS_WAVE(10) GF_RMS_WAVE(8) GF_RMS_WAVE(3)
#undef GF_RMS_WAVE
#undef GF_RMS_WAVE_ROPE
    hipLaunchKernelGGL(rmsnorm_rope_kernel, dim3((unsigned)rows), dim3(ROW_THREADS), 0, (hipStream_t)stream,
                       (u16*)x, (const u16*)weight, cos_tab, sin_tab, (int)dim, (int)head_dim, (long)x_stride, eps);
    GF_CHECK_LAUNCH("gf_rmsnorm_rope");
    return GF_OK;
}
