// gf_vae.hip — data-movement / normalisation kernels of the Wan 3-D causal VAE decoder
// (reference diffsynth/models/wan_video_vae.py: CausalConv3d 33-52, RMS_norm 55-70, Resample 82-174,
// ResidualBlock 267-301, AttentionBlock 304-342, Decoder3d 736-838, VideoVAE_.decode 1011-1034,
// WanVideoVAE.tiled_decode 1103-1152).
//
// Layout: decoder activations are channels-last [T, H, W, C] bf16, so a pixel's channel vector is one
// contiguous row.  Every convolution is  patch-gather (this file)  +  the MFMA GEMM of gf_gemm.hip with the
// weight pre-permuted to [Cout, (dt, dy, dx, cin)]:
//   * causal 3x3x3 / 3x1x1 convs read their 2-frame temporal halo from the per-layer feature cache
//     (the reference's feat_cache with CACHE_T = 2; an all-zero cache == "no cache yet" zero padding);
//   * the nearest-exact 2x upsample of Resample is folded into the gather of the following 3x3 conv.
// All kernels are HBM-bound: 16 bytes per lane, one pixel row per wave where a reduction is needed.
#include "gf_common.h"

namespace {

constexpr int VT = 256;
static inline unsigned vgrid(long items) {
    long b = (items + VT - 1) / VT;
    if (b < 1) b = 1;
    if (b > 256 * 16) b = 256 * 16;
    return (unsigned)b;
}

// z (NCTHW slice, arbitrary strides) -> channels-last [T,H,W,cpad]: bf16(bf16(z / inv_std) + mean)
// (VideoVAE_.decode VAE:1014-1020 with scale = [mean, 1/std] cast to bf16), channels >= C zero.
__global__ __launch_bounds__(VT) void prep_latent_kernel(const u16* __restrict__ z, long sc, long st, long sy, long sx,
                                                         const u16* __restrict__ mean, const u16* __restrict__ inv_std,
                                                         u16* __restrict__ out, int C, int T, int H, int W, int cpad) {
    const long total = (long)T * H * W * cpad;
    const long stride = (long)gridDim.x * VT;
    for (long i = (long)blockIdx.x * VT + threadIdx.x; i < total; i += stride) {
        const int c = (int)(i % cpad);
        long p = i / cpad;
        const int x = (int)(p % W);
        p /= W;
        const int y = (int)(p % H);
        const int t = (int)(p / H);
        u16 v = 0;
        if (c < C) {
            const float zz = bf2f(z[c * sc + t * st + y * sy + x * sx]);
            v = f2bf(rbf(zz / bf2f(inv_std[c])) + bf2f(mean[c]));
        }
        out[i] = v;
    }
}

// Patch gather for conv-as-GEMM.  out[(j, Y, X), ((dt*ks + dy)*ks + dx)*C + c] with zero fill for spatial
// padding, temporal history from `cache` (2 frames) and the K padding columns.  Output frame j is the
// causal conv evaluated at input frame t = t_off + j*t_stride (taps t-kt+1 .. t).
// mode 0: stride 1, symmetric zero padding ks/2.
// mode 1: the conv runs on the nearest-2x-upsampled image (source pixel = (Y', X') >> 1)  [Resample upsample*].
// mode 2: stride 2 with ZeroPad2d((0,1,0,1)) — taps (2Y+dy, 2X+dx), zero past the bottom/right edge
//         [Resample downsample*, VAE:101-112].
// One thread per 16-byte chunk (8 channels).
__global__ __launch_bounds__(VT) void im2col_kernel(const u16* __restrict__ src, const u16* __restrict__ cache,
                                                    u16* __restrict__ out, int T_out, int H, int W, int C, int kt,
                                                    int ks, int mode, int t_stride, int t_off, int kpad) {
    const int Ho = mode == 1 ? 2 * H : (mode == 2 ? H / 2 : H);
    const int Wo = mode == 1 ? 2 * W : (mode == 2 ? W / 2 : W);
    const int cpc = C >> 3;                  // chunks per tap
    const int kchunks = kpad >> 3;           // chunks per output row
    const int taps = kt * ks * ks;
    const long rows = (long)T_out * Ho * Wo;
    const long total = rows * kchunks;
    const long frame = (long)H * W * C;
    const int half = ks >> 1;
    const long stride = (long)gridDim.x * VT;
    for (long i = (long)blockIdx.x * VT + threadIdx.x; i < total; i += stride) {
        const int kc = (int)(i % kchunks);
        const long row = i / kchunks;
        u16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
        const int tap = kc / cpc;
        if (tap < taps) {
            const int c8 = kc - tap * cpc;
            const int dx = tap % ks, dy = (tap / ks) % ks, dt = tap / (ks * ks);
            const int X = (int)(row % Wo);
            const int Y = (int)((row / Wo) % Ho);
            const int j = (int)(row / ((long)Wo * Ho));
            int sy, sx;
            bool ok;
            if (mode == 2) {
                sy = 2 * Y + dy;
                sx = 2 * X + dx;
                ok = sy < H && sx < W;
            } else {
                const int yy = Y + dy - half, xx = X + dx - half;
                ok = yy >= 0 && yy < Ho && xx >= 0 && xx < Wo;
                sy = mode == 1 ? (yy >> 1) : yy;
                sx = mode == 1 ? (xx >> 1) : xx;
            }
            if (ok) {
                const int f = t_off + j * t_stride - (kt - 1) + dt;  // causal: taps reach kt-1 frames into the past
                const u16* base = (f >= 0) ? (src + (long)f * frame) : (cache + (long)(2 + f) * frame);
                v = *reinterpret_cast<const u16x8*>(base + ((long)sy * W + sx) * C + (c8 << 3));
            }
        }
        *reinterpret_cast<u16x8*>(out + row * kpad + ((long)kc << 3)) = v;
    }
}

// Encoder tail (VideoVAE_.encode VAE:1002-1010): mu = first C channels of the conv1 output; out = bf16(bf16(mu - mean) * inv_std),
// channels-last [rows, C] -> [rows, C].
__global__ __launch_bounds__(VT) void finish_latent_kernel(const u16* __restrict__ x, long ldx, const u16* __restrict__ mean,
                                                           const u16* __restrict__ inv_std, u16* __restrict__ out,
                                                           long rows, int C) {
    const long total = rows * C;
    const long stride = (long)gridDim.x * VT;
    for (long i = (long)blockIdx.x * VT + threadIdx.x; i < total; i += stride) {
        const int c = (int)(i % C);
        const long r = i / C;
        out[i] = f2bf(rbf(bf2f(x[r * ldx + c]) - bf2f(mean[c])) * bf2f(inv_std[c]));
    }
}

// RMS_norm (VAE:55-70): F.normalize(x, dim=channel) * sqrt(C) * gamma, then optional SiLU; every eager op of
// the reference rounds to bf16 and so do we.  LPR lanes per pixel row (16 bytes per lane, C <= 8*LPR): 64/LPR rows per wave,
// so the decoder's widest images (C = 96: 12 chunks) keep 48 of 64 lanes busy instead of 12.  The row sum is the upper part
// of the same descending butterfly at every LPR (the stages it skips would only add zeros): results do not depend on LPR.
template <int LPR>
__global__ __launch_bounds__(VT) void rmsnorm_silu_kernel(const u16* __restrict__ x, const u16* __restrict__ gamma,
                                                          u16* __restrict__ out, long rows, int C, float scale,
                                                          int silu) {
    constexpr int RPW = 64 / LPR;                 // rows per wave
    const int lane = threadIdx.x & 63;
    const int sub = lane & (LPR - 1);
    const long row0 = ((long)blockIdx.x * (VT / 64) + (threadIdx.x >> 6)) * RPW + lane / LPR;
    const long rstride = (long)gridDim.x * (VT / 64) * RPW;
    const int nch = C >> 3;
    u16x8 g8 = {0, 0, 0, 0, 0, 0, 0, 0};
    if (sub < nch) g8 = *reinterpret_cast<const u16x8*>(gamma + (sub << 3));
    // the loop bound is wave-uniform (the shuffles below need every lane): rows past the end are computed on zeros, not stored.
    // UNR row groups per iteration: their loads are all in flight before the first is used (one 16-byte load per lane and iteration
    // left ~32 KB in flight per CU: 4 TB/s at HBM latency, whatever the arithmetic cost).
    constexpr int UNR = 4;
    for (long rb = row0 - lane / LPR; rb < rows; rb += UNR * rstride) {
        u16x8 v[UNR];
        bool live[UNR];
        long rowi[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            rowi[u] = rb + u * rstride + lane / LPR;
            live[u] = sub < nch && rowi[u] < rows;
            v[u] = u16x8{0, 0, 0, 0, 0, 0, 0, 0};
            if (live[u]) v[u] = *reinterpret_cast<const u16x8*>(x + rowi[u] * C + (sub << 3));
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            if (rb + u * rstride >= rows) break;          // wave-uniform
            float s = 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float f = bf2f(v[u][j]);
                s += f * f;
            }
#pragma unroll
            for (int o = LPR / 2; o > 0; o >>= 1) s += __shfl_xor(s, o);
            const float nrm = fmaxf(rbf(sqrtf(s)), 1e-12f);  // x.norm(2, dim).clamp_min(eps), bf16 tensor
            // x / norm: ONE correctly rounded reciprocal per row, then a multiply per value.  Bit-identical to the division after the
            // bf16 rounding that follows: the quotient of two bf16 numbers (8-bit significands mx / mn) is either exactly a bf16 number
            // or at least 2^-17 (relative) away from every bf16 rounding boundary — mx 2^9 - k mn is a non-zero integer — while x * (1/n)
            // is within 2^-23 of it; an exact tie cannot occur (it would need a 9-bit quotient of two 8-bit integers' ratio).  The
            // per-value IEEE division (10 instructions) and expf + division of SiLU made this kernel VALU-bound at 4.4 TB/s.
            const float rinv = 1.0f / nrm;
            if (live[u]) {
                u32x4 o;
                const u32x4 vv = __builtin_bit_cast(u32x4, v[u]), gg = __builtin_bit_cast(u32x4, g8);
#pragma unroll
                for (int p = 0; p < 4; ++p) {                  // values in pairs: one v_cvt_pk_bf16_f32 per rounding and pair
                    float y0 = __uint_as_float(vv[p] << 16) * rinv, y1 = __uint_as_float(vv[p] & 0xffff0000u) * rinv;      // x / norm
                    unsigned r = pack2bf(y0, y1);
                    y0 = __uint_as_float(r << 16) * scale, y1 = __uint_as_float(r & 0xffff0000u) * scale;                    // * dim**0.5
                    r = pack2bf(y0, y1);
                    y0 = __uint_as_float(r << 16) * __uint_as_float(gg[p] << 16);                                            // * gamma (+ bias 0.)
                    y1 = __uint_as_float(r & 0xffff0000u) * __uint_as_float(gg[p] & 0xffff0000u);
                    r = pack2bf(y0, y1);
                    if (silu) {   // x * sigmoid(x) on the bf16 values by v_exp_f32 + v_rcp_f32 (about 1 fp32 ulp each) instead of expf and an
                                  // IEEE division: a few fp32 ulp before the bf16 rounding, so a value that sits within that distance of a
                                  // bf16 rounding boundary could come out ONE bf16 ulp away from torch's F.silu (measured on an MI355X: 0 of
                                  // 1.57 M values differ at C = 96 / 128 / 384: tests/test_vae.py::test_hip_vae_silu_is_within_one_ulp_of_exact_division);
                                  // only the x * (1 / n) step above is exact by construction.  rmsnorm_silu3_kernel shares this code.
                        y0 = __uint_as_float(r << 16), y1 = __uint_as_float(r & 0xffff0000u);
                        const float t0 = __builtin_amdgcn_exp2f(y0 * -1.4426950408889634f), t1 = __builtin_amdgcn_exp2f(y1 * -1.4426950408889634f);
                        y0 = y0 * __builtin_amdgcn_rcpf(1.0f + t0), y1 = y1 * __builtin_amdgcn_rcpf(1.0f + t1);
                        r = pack2bf(y0, y1);
                    }
                    o[p] = r;
                }
                *reinterpret_cast<u32x4*>(out + rowi[u] * C + (sub << 3)) = o;
            }
        }
    }
}

// The same for C = 96, 192, 384 (three quarters of a power of two: a quarter of the lanes above load nothing, which caps the kernel at
// ~4.9 TB/s): LP lanes per pixel with THREE 16-byte chunks each (sub, sub + LP, sub + 2 LP), every lane live.  The row sum is the same
// tree as above — its two top butterfly stages pair chunk c with c ^ 2 LP and c ^ LP, which are now additions inside the lane
// ((p[sub] + p[sub + 2 LP]) + (p[sub + LP] + 0)) — so the results are bit-identical to rmsnorm_silu_kernel<4 LP>.
// PAD: the output is the interior of a zero-bordered buffer [T][H + 2][W + 2][C] (gf_conv_a4.hip): pixel (t, y, x) of the contiguous
// input goes to row (t (H + 2) + y) (W + 2) + x behind `out` (which points at the interior's first pixel); hw = H W, w = W.
template <int LP, bool PAD = false>
__global__ __launch_bounds__(VT) void rmsnorm_silu3_kernel(const u16* __restrict__ x, const u16* __restrict__ gamma,
                                                           u16* __restrict__ out, long rows, float scale, int silu, int hw = 0, int w = 0) {
    constexpr int RPW = 64 / LP, C = 24 * LP;
    const int lane = threadIdx.x & 63;
    const int sub = lane & (LP - 1);
    const long row0 = ((long)blockIdx.x * (VT / 64) + (threadIdx.x >> 6)) * RPW;
    const long rstride = (long)gridDim.x * (VT / 64) * RPW;
    u32x4 gg[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) gg[k] = *reinterpret_cast<const u32x4*>(gamma + ((sub + k * LP) << 3));
    constexpr int UNR = 2;
    for (long rb = row0; rb < rows; rb += UNR * rstride) {        // wave-uniform bound
        u16x8 v[UNR][3];
        long rowi[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            rowi[u] = rb + u * rstride + lane / LP;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                v[u][k] = u16x8{0, 0, 0, 0, 0, 0, 0, 0};
                if (rowi[u] < rows) v[u][k] = *reinterpret_cast<const u16x8*>(x + rowi[u] * C + ((sub + k * LP) << 3));
            }
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            if (rb + u * rstride >= rows) break;
            float ps[3];
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                float t = 0.f;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float f = bf2f(v[u][k][j]);
                    t += f * f;
                }
                ps[k] = t;
            }
            float s = (ps[0] + ps[2]) + (ps[1] + 0.f);
#pragma unroll
            for (int o = LP / 2; o > 0; o >>= 1) s += __shfl_xor(s, o);
            const float nrm = fmaxf(rbf(sqrtf(s)), 1e-12f);
            const float rinv = 1.0f / nrm;
            if (rowi[u] < rows) {
                long orow = rowi[u];
                if constexpr (PAD) {
                    const int t = (int)(rowi[u] / hw), rem = (int)(rowi[u] - (long)t * hw);
                    const int y = rem / w, xx = rem - y * w;
                    orow = ((long)t * (hw / w + 2) + y) * (w + 2) + xx;
                }
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    u32x4 o;
                    const u32x4 vv = __builtin_bit_cast(u32x4, v[u][k]);
#pragma unroll
                    for (int p = 0; p < 4; ++p) {
                        float y0 = __uint_as_float(vv[p] << 16) * rinv, y1 = __uint_as_float(vv[p] & 0xffff0000u) * rinv;
                        unsigned r = pack2bf(y0, y1);
                        y0 = __uint_as_float(r << 16) * scale, y1 = __uint_as_float(r & 0xffff0000u) * scale;
                        r = pack2bf(y0, y1);
                        y0 = __uint_as_float(r << 16) * __uint_as_float(gg[k][p] << 16);
                        y1 = __uint_as_float(r & 0xffff0000u) * __uint_as_float(gg[k][p] & 0xffff0000u);
                        r = pack2bf(y0, y1);
                        if (silu) {
                            y0 = __uint_as_float(r << 16), y1 = __uint_as_float(r & 0xffff0000u);
                            const float t0 = __builtin_amdgcn_exp2f(y0 * -1.4426950408889634f), t1 = __builtin_amdgcn_exp2f(y1 * -1.4426950408889634f);
                            y0 = y0 * __builtin_amdgcn_rcpf(1.0f + t0), y1 = y1 * __builtin_amdgcn_rcpf(1.0f + t1);
                            r = pack2bf(y0, y1);
                        }
                        o[p] = r;
                    }
                    *reinterpret_cast<u32x4*>(out + orow * C + ((sub + k * LP) << 3)) = o;
                }
            }
        }
    }
}

// out[r, :ncols] = softmax(bf16(x[r, :] * scale + bias[r, :])) over the first nvalid columns (columns >= nvalid get
// probability 0 = the reference's masked_fill(finfo.min)), fp32 math, bf16 out; out[r, ncols:ldo] = 0.
// One workgroup per row.  Used by the VAE AttentionBlock (VAE:326-333; no bias) and by the umT5 attention
// (wan_video_text_encoder.py:72-84: scores + position bias (+ key mask), softmax in fp32, no scaling).
__global__ __launch_bounds__(VT) void softmax_rows_kernel(const u16* __restrict__ x, long ldx, const u16* __restrict__ bias,
                                                          long ldb, u16* __restrict__ out, long ldo, int ncols, int nvalid,
                                                          float scale) {
    __shared__ float red[VT / 64];
    const long row = blockIdx.x;
    const u16* xr = x + row * ldx;
    const u16* br = bias ? bias + row * ldb : nullptr;
    auto val = [&](int c) {
        float v = bf2f(xr[c]) * scale;
        if (br) v = rbf(v + bf2f(br[c]));   // attn = scores + attn_bias, a bf16 tensor
        return v;
    };
    float mx = -INFINITY;
    for (int c = threadIdx.x; c < nvalid; c += VT) mx = fmaxf(mx, val(c));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    float s = 0.f;
    for (int c = threadIdx.x; c < nvalid; c += VT) s += expf(val(c) - mx);
    const float tot = block_sum<VT>(s, red);
    const float inv = 1.0f / tot;
    u16* orow = out + row * ldo;
    for (int c = threadIdx.x; c < (int)ldo; c += VT)
        orow[c] = c < nvalid ? f2bf(expf(val(c) - mx) * inv) : (u16)0;
}

// out[r * ldo] = bf16(-max_c x[r, c]) — the per-row offset of the AttentionBlock's score GEMM (gf_rowmax_neg_bf16 below).  One
// workgroup per row; the value is exact (a maximum of bf16 numbers, negated).
__global__ __launch_bounds__(VT) void rowmax_neg_kernel(const u16* __restrict__ x, long ldx, u16* __restrict__ out, long ldo, int ncols) {
    __shared__ float red[VT / 64];
    const long row = blockIdx.x;
    const u16* xr = x + row * ldx;
    float mx = -INFINITY;
    for (int c = threadIdx.x; c < ncols; c += VT) mx = fmaxf(mx, bf2f(xr[c]));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
    __syncthreads();
    if (threadIdx.x == 0) out[row * ldo] = f2bf(-fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3])));
}

// dst[c, r] = src[r, c] for r < R (zero for R <= r < rpad); 32x32 tiles through LDS.
__global__ __launch_bounds__(VT) void transpose_pad_kernel(const u16* __restrict__ src, long lds_, u16* __restrict__ dst,
                                                           int R, int C, int rpad, long bsrc, long bdst) {
    src += (long)blockIdx.z * bsrc;              // gf_transpose_pad_batched: matrix blockIdx.z
    dst += (long)blockIdx.z * bdst;
    __shared__ u16 tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    const int r0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
    for (int j = ty; j < 32; j += 8) {
        const int r = r0 + j, c = c0 + tx;
        tile[j][tx] = (r < R && c < C) ? src[(long)r * lds_ + c] : (u16)0;
    }
    __syncthreads();
    for (int j = ty; j < 32; j += 8) {
        const int c = c0 + j, r = r0 + tx;
        if (c < C && r < rpad) dst[(long)c * rpad + r] = tile[tx][j];
    }
}

// The same transpose in 64 x 64 tiles with 16-byte accesses on both sides (the 32 x 32 kernel above moves 2 bytes per lane: 0.37 ms
// for the [32760, 5120] operands of the training step's weight-gradient GEMMs against 0.11 ms for the bytes).  Needs C % 8 == 0,
// rpad % 8 == 0 and 16-byte aligned rows on both sides.
__global__ __launch_bounds__(VT) void transpose_pad64_kernel(const u16* __restrict__ src, long lds_, u16* __restrict__ dst, int R, int C,
                                                             int rpad, long bsrc, long bdst) {
    src += (long)blockIdx.z * bsrc;
    dst += (long)blockIdx.z * bdst;
    __shared__ u16 tile[64][64 + 2];             // 33-dword rows: the column reads below walk banks
    const int t = threadIdx.x;
    const int r0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int row = (t >> 3) + 32 * it, ch = t & 7;
        const int r = r0 + row, c = c0 + 8 * ch;
        u16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
        if (r < R && c < C) v = *reinterpret_cast<const u16x8*>(src + (long)r * lds_ + c);
#pragma unroll
        for (int e = 0; e < 8; ++e) tile[row][8 * ch + e] = v[e];
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int oc = (t >> 3) + 32 * it, ch = t & 7;      // output row = source column c0 + oc; 8 source rows r0 + 8 ch ..
        const int c = c0 + oc, r = r0 + 8 * ch;
        if (c < C && r < rpad) {
            u16x8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = tile[8 * ch + e][oc];
            *reinterpret_cast<u16x8*>(dst + (long)c * rpad + r) = o;
        }
    }
}

// nn.Upsample(scale_factor=(2, 2), mode="nearest-exact") (VAE:82-96) of a contiguous [T, H, W, C] activation, written into the
// INTERIOR of a zero-bordered buffer [T][2 H + 2][2 W + 2][C] (the input layout of gf_conv3d_padded_bf16 with kt = 1): source pixel
// (t, y, x) goes to the four pixels (2 y + a, 2 x + b).  One thread per 16-byte chunk of a source pixel.
__global__ __launch_bounds__(VT) void upsample2x_padded_kernel(const u16* __restrict__ x, u16* __restrict__ out, long pixels, int H, int W,
                                                               int cpp /* 16-byte chunks per pixel */) {
    const long total = pixels * cpp;
    const long stride = (long)gridDim.x * VT;
    const int Wp = 2 * W + 2, Hp = 2 * H + 2;
    for (long i = (long)blockIdx.x * VT + threadIdx.x; i < total; i += stride) {
        const long pix = i / cpp;
        const int ch = (int)(i - pix * cpp);
        const int xx = (int)(pix % W);
        const int y = (int)((pix / W) % H);
        const long t = pix / ((long)W * H);
        const u16x8 v = __builtin_nontemporal_load(reinterpret_cast<const u16x8*>(x + pix * (cpp * 8L) + ch * 8));
        u16* o = out + ((t * Hp + 2 * y) * Wp + 2 * xx) * (cpp * 8L) + ch * 8;
        __builtin_nontemporal_store(v, reinterpret_cast<u16x8*>(o));
        __builtin_nontemporal_store(v, reinterpret_cast<u16x8*>(o + cpp * 8L));
        __builtin_nontemporal_store(v, reinterpret_cast<u16x8*>(o + (long)Wp * cpp * 8L));
        __builtin_nontemporal_store(v, reinterpret_cast<u16x8*>(o + (long)(Wp + 1) * cpp * 8L));
    }
}

// Tile blend accumulate (WanVideoVAE.tiled_decode VAE:1128-1150, bf16 accumulators):
//   values[c,t,Y,X] += tile[t,y,x,c] * mask(y,x);  weight[Y,X] += mask(y,x)
// mask = min(ramp_h(y), ramp_w(x)) with ramps (i+1)/border on non-boundary sides (build_mask VAE:1081-1100).
__global__ __launch_bounds__(VT) void tile_blend_kernel(u16* __restrict__ values, u16* __restrict__ weight,
                                                        const u16* __restrict__ tile, int nch, int T, int th, int tw,
                                                        int tc, int H, int W, int y0, int x0, int top, int bottom,
                                                        int left, int right, int bh, int bw) {
    const long total = (long)T * th * tw;
    const long stride = (long)gridDim.x * VT;
    for (long i = (long)blockIdx.x * VT + threadIdx.x; i < total; i += stride) {
        const int x = (int)(i % tw);
        const int y = (int)((i / tw) % th);
        const int t = (int)(i / ((long)tw * th));
        float mh = 1.f, mw = 1.f;
        if (!top && y < bh) mh = (float)(y + 1) / (float)bh;
        if (!bottom && y >= th - bh) mh = (float)(th - y) / (float)bh;
        if (!left && x < bw) mw = (float)(x + 1) / (float)bw;
        if (!right && x >= tw - bw) mw = (float)(tw - x) / (float)bw;
        const float m = rbf(fminf(mh, mw));  // mask.to(bf16)
        const long pix = (long)(y0 + y) * W + (x0 + x);
        for (int c = 0; c < nch; ++c) {
            u16* vp = values + ((long)c * T + t) * H * W + pix;
            *vp = f2bf(bf2f(*vp) + rbf(bf2f(tile[i * tc + c]) * m));
        }
        if (t == 0) weight[pix] = f2bf(bf2f(weight[pix]) + m);
    }
}

// values = clamp(values / weight, -1, 1)   (VAE:1149-1151)
__global__ __launch_bounds__(VT) void tile_finalize_kernel(u16* __restrict__ values, const u16* __restrict__ weight,
                                                           long planes, long hw, int clamp1) {
    const long total = planes * hw;
    const long stride = (long)gridDim.x * VT;
    for (long i = (long)blockIdx.x * VT + threadIdx.x; i < total; i += stride) {
        const float v = rbf(bf2f(values[i]) / bf2f(weight[i % hw]));
        values[i] = f2bf(clamp1 ? fminf(fmaxf(v, -1.f), 1.f) : v);
    }
}

}  // namespace

extern "C" GF_API int gf_vae_prep_latent(const void* z, int64_t sc, int64_t st, int64_t sy, int64_t sx,
                                         const void* mean, const void* inv_std, void* out, int64_t C, int64_t T,
                                         int64_t H, int64_t W, int64_t cpad, void* stream) {
    GF_CHECK_ARG(z && mean && inv_std && out && C > 0 && cpad >= C && T > 0 && H > 0 && W > 0,
                 "gf_vae_prep_latent: bad arguments");
    hipLaunchKernelGGL(prep_latent_kernel, dim3(vgrid(T * H * W * cpad)), dim3(VT), 0, (hipStream_t)stream,
                       (const u16*)z, (long)sc, (long)st, (long)sy, (long)sx, (const u16*)mean, (const u16*)inv_std,
                       (u16*)out, (int)C, (int)T, (int)H, (int)W, (int)cpad);
    GF_CHECK_LAUNCH("gf_vae_prep_latent");
    return GF_OK;
}

extern "C" GF_API int gf_vae_im2col(const void* src, const void* cache, void* out, int64_t T_out, int64_t H, int64_t W,
                                    int64_t C, int64_t kt, int64_t ks, int mode, int64_t t_stride, int64_t t_off,
                                    int64_t kpad, void* stream) {
    GF_CHECK_ARG(src && out && T_out > 0 && H > 0 && W > 0, "gf_vae_im2col: bad arguments");
    GF_CHECK_ARG(C > 0 && C % 8 == 0 && kpad % 8 == 0 && kpad >= kt * ks * ks * C,
                 "gf_vae_im2col: C=%ld must be a multiple of 8 and kpad=%ld >= taps*C", (long)C, (long)kpad);
    GF_CHECK_ARG((kt == 1 || kt == 3) && (ks == 1 || ks == 3), "gf_vae_im2col: kernel must be 1 or 3 per axis");
    GF_CHECK_ARG(mode >= 0 && mode <= 2 && t_stride >= 1 && t_off >= 0, "gf_vae_im2col: bad mode / temporal stride");
    GF_CHECK_ARG(mode != 2 || (H % 2 == 0 && W % 2 == 0 && ks == 3), "gf_vae_im2col: downsample mode needs even H, W and ks=3");
    GF_CHECK_ARG(kt == 1 || cache, "gf_vae_im2col: a temporal kernel needs the 2-frame cache (zeros = no history)");
    GF_CHECK_ARG(gf_aligned16(src) && gf_aligned16(out) && (!cache || gf_aligned16(cache)),
                 "gf_vae_im2col: 16-byte alignment required");
    const long rows = mode == 1 ? T_out * H * W * 4 : (mode == 2 ? T_out * (H / 2) * (W / 2) : T_out * H * W);
    hipLaunchKernelGGL(im2col_kernel, dim3(vgrid(rows * (kpad / 8))), dim3(VT), 0, (hipStream_t)stream,
                       (const u16*)src, (const u16*)cache, (u16*)out, (int)T_out, (int)H, (int)W, (int)C, (int)kt, (int)ks,
                       mode, (int)t_stride, (int)t_off, (int)kpad);
    GF_CHECK_LAUNCH("gf_vae_im2col");
    return GF_OK;
}

extern "C" GF_API int gf_vae_finish_latent(const void* x, int64_t ldx, const void* mean, const void* inv_std, void* out,
                                           int64_t rows, int64_t C, void* stream) {
    GF_CHECK_ARG(x && mean && inv_std && out && rows >= 0 && C > 0 && ldx >= C, "gf_vae_finish_latent: bad arguments");
    if (rows == 0) return GF_OK;
    hipLaunchKernelGGL(finish_latent_kernel, dim3(vgrid(rows * C)), dim3(VT), 0, (hipStream_t)stream, (const u16*)x,
                       (long)ldx, (const u16*)mean, (const u16*)inv_std, (u16*)out, (long)rows, (int)C);
    GF_CHECK_LAUNCH("gf_vae_finish_latent");
    return GF_OK;
}

extern "C" GF_API int gf_vae_rmsnorm_silu(const void* x, const void* gamma, void* out, int64_t rows, int64_t C,
                                          int silu, void* stream) {
    GF_CHECK_ARG(x && gamma && out && rows >= 0, "gf_vae_rmsnorm_silu: bad arguments");
    GF_CHECK_ARG(C > 0 && C % 8 == 0 && C <= 512, "gf_vae_rmsnorm_silu: C=%ld must be a multiple of 8 and <= 512", (long)C);
    GF_CHECK_ARG(gf_aligned16(x) && gf_aligned16(out) && gf_aligned16(gamma), "gf_vae_rmsnorm_silu: alignment");
    if (rows == 0) return GF_OK;
#define GF_RMS_LAUNCH(LPR)                                                                                                  \
    hipLaunchKernelGGL(rmsnorm_silu_kernel<LPR>, dim3(vgrid(rows * LPR)), dim3(VT), 0, (hipStream_t)stream, (const u16*)x,  \
                       (const u16*)gamma, (u16*)out, (long)rows, (int)C, sqrtf((float)C), silu ? 1 : 0)
#define GF_RMS3_LAUNCH(LP)                                                                                                   \
    hipLaunchKernelGGL(rmsnorm_silu3_kernel<LP>, dim3(vgrid(rows * LP)), dim3(VT), 0, (hipStream_t)stream, (const u16*)x,       \
                       (const u16*)gamma, (u16*)out, (long)rows, sqrtf((float)C), silu ? 1 : 0)
    const bool rms3 = gf_options().vae_rms3.load(std::memory_order_relaxed) != 0;
    if (rms3 && C == 96) GF_RMS3_LAUNCH(4);
    else if (rms3 && C == 192) GF_RMS3_LAUNCH(8);
    else if (rms3 && C == 384) GF_RMS3_LAUNCH(16);
    else if (C <= 64) GF_RMS_LAUNCH(8);
    else if (C <= 128) GF_RMS_LAUNCH(16);
    else if (C <= 256) GF_RMS_LAUNCH(32);
    else GF_RMS_LAUNCH(64);
#undef GF_RMS_LAUNCH
#undef GF_RMS3_LAUNCH
    GF_CHECK_LAUNCH("gf_vae_rmsnorm_silu");
    return GF_OK;
}

// RMS_norm (+SiLU) of a contiguous [T, H, W, C] activation written into the INTERIOR of a zero-bordered buffer [T][H + 2][W + 2][C]:
// `out_interior` = the address of padded pixel (frame 0, y = 1, x = 1), i.e. of the first real pixel.  Same arithmetic, same kernel
// as gf_vae_rmsnorm_silu (C = 192 / 384: the three-chunk kernel); the borders are the caller's (zero them once).
extern "C" GF_API int gf_vae_rmsnorm_silu_padded(const void* x, const void* gamma, void* out_interior, int64_t T, int64_t H, int64_t W,
                                                 int64_t C, int silu, void* stream) {
    GF_CHECK_ARG(x && gamma && out_interior && T >= 0 && H > 0 && W > 0, "gf_vae_rmsnorm_silu_padded: bad arguments");
    if (!(C == 192 || C == 384)) {
        gf_set_error("gf_vae_rmsnorm_silu_padded: C=%ld (192 or 384: the levels whose convolutions read the padded layout)", (long)C);
        return GF_ERR_UNSUPPORTED;
    }
    GF_CHECK_ARG(gf_aligned16(x) && gf_aligned16(out_interior) && gf_aligned16(gamma), "gf_vae_rmsnorm_silu_padded: alignment");
    GF_CHECK_ARG(T * H * W < (1LL << 31) && H * W < (1LL << 30), "gf_vae_rmsnorm_silu_padded: too many pixels");
    const long rows = T * H * W;
    if (rows == 0) return GF_OK;
    if (C == 192)
        hipLaunchKernelGGL((rmsnorm_silu3_kernel<8, true>), dim3(vgrid(rows * 8)), dim3(VT), 0, (hipStream_t)stream, (const u16*)x,
                           (const u16*)gamma, (u16*)out_interior, rows, sqrtf((float)C), silu ? 1 : 0, (int)(H * W), (int)W);
    else
        hipLaunchKernelGGL((rmsnorm_silu3_kernel<16, true>), dim3(vgrid(rows * 16)), dim3(VT), 0, (hipStream_t)stream, (const u16*)x,
                           (const u16*)gamma, (u16*)out_interior, rows, sqrtf((float)C), silu ? 1 : 0, (int)(H * W), (int)W);
    GF_CHECK_LAUNCH("gf_vae_rmsnorm_silu_padded");
    return GF_OK;
}

// out_interior = &xp[0][1][1][0] of the zero-bordered [T][2 H + 2][2 W + 2][C] buffer (borders: the caller zeroes them once)
extern "C" GF_API int gf_vae_upsample2x_padded(const void* x, void* out_interior, int64_t T, int64_t H, int64_t W, int64_t C, void* stream) {
    GF_CHECK_ARG(x && out_interior && T >= 0 && H > 0 && W > 0 && C > 0 && C % 8 == 0, "gf_vae_upsample2x_padded: bad arguments (C a multiple of 8)");
    GF_CHECK_ARG(gf_aligned16(x) && gf_aligned16(out_interior), "gf_vae_upsample2x_padded: alignment");
    const long pixels = T * H * W;
    if (pixels == 0) return GF_OK;
    hipLaunchKernelGGL(upsample2x_padded_kernel, dim3(vgrid(pixels * (C / 8))), dim3(VT), 0, (hipStream_t)stream, (const u16*)x,
                       (u16*)out_interior, pixels, (int)H, (int)W, (int)(C / 8));
    GF_CHECK_LAUNCH("gf_vae_upsample2x_padded");
    return GF_OK;
}

extern "C" GF_API int gf_softmax_rows(const void* x, int64_t ldx, const void* bias, int64_t ldb, void* out, int64_t ldo,
                                      int64_t rows, int64_t ncols, int64_t nvalid, float scale, void* stream) {
    GF_CHECK_ARG(x && out && rows >= 0 && ncols > 0 && ldx >= ncols && ldo >= ncols, "gf_softmax_rows: bad arguments");
    GF_CHECK_ARG(nvalid > 0 && nvalid <= ncols && (!bias || ldb >= ncols), "gf_softmax_rows: bad nvalid / bias stride");
    if (rows == 0) return GF_OK;
    hipLaunchKernelGGL(softmax_rows_kernel, dim3((unsigned)rows), dim3(VT), 0, (hipStream_t)stream, (const u16*)x,
                       (long)ldx, (const u16*)bias, (long)ldb, (u16*)out, (long)ldo, (int)ncols, (int)nvalid, scale);
    GF_CHECK_LAUNCH("gf_softmax_rows");
    return GF_OK;
}

// gf_rowmax_neg_bf16 — out[r * ldo] = -max_c x[r, c] (bf16, exact).  The VAE AttentionBlock (VAE:304-342) takes its softmax over
// scores that went through a bf16 GEMM output: rounding the RAW scores costs 2^-9 of their magnitude, a 1 % change of a softmax weight
// at logit 5, which F.scaled_dot_product_attention (fp32 scores) does not pay.  The softmax only needs s - (row constant): this
// kernel writes minus the row maximum of a first, coarse score GEMM into an extra K column of the Q operand (k' carries a 1
// there), and the second GEMM then accumulates q.k - max in fp32 and rounds a number that is SMALL exactly where the softmax weight is
// large (goal_force_amd/vae.py::_attention).
extern "C" GF_API int gf_rowmax_neg_bf16(const void* x, int64_t ldx, void* out, int64_t ldo, int64_t rows, int64_t ncols, void* stream) {
    GF_CHECK_ARG(x && out && rows >= 0 && ncols > 0 && ldx >= ncols && ldo >= 1, "gf_rowmax_neg_bf16: bad arguments");
    if (rows == 0) return GF_OK;
    hipLaunchKernelGGL(rowmax_neg_kernel, dim3((unsigned)rows), dim3(VT), 0, (hipStream_t)stream, (const u16*)x, (long)ldx, (u16*)out,
                       (long)ldo, (int)ncols);
    GF_CHECK_LAUNCH("gf_rowmax_neg_bf16");
    return GF_OK;
}

static int transpose_pad_launch(const char* fn, const void* src, int64_t ld_src, int64_t stride_src, void* dst, int64_t stride_dst, int64_t R,
                                int64_t C, int64_t rpad, int64_t batch, void* stream) {
    if (C % 8 == 0 && rpad % 8 == 0 && ld_src % 8 == 0 && stride_src % 8 == 0 && stride_dst % 8 == 0 && gf_aligned16(src) && gf_aligned16(dst) &&
        VT == 256)
        hipLaunchKernelGGL(transpose_pad64_kernel, dim3((unsigned)((rpad + 63) / 64), (unsigned)((C + 63) / 64), (unsigned)batch), dim3(VT), 0,
                           (hipStream_t)stream, (const u16*)src, (long)ld_src, (u16*)dst, (int)R, (int)C, (int)rpad, (long)stride_src,
                           (long)stride_dst);
    else
        hipLaunchKernelGGL(transpose_pad_kernel, dim3((unsigned)((rpad + 31) / 32), (unsigned)((C + 31) / 32), (unsigned)batch), dim3(VT), 0,
                           (hipStream_t)stream, (const u16*)src, (long)ld_src, (u16*)dst, (int)R, (int)C, (int)rpad, (long)stride_src,
                           (long)stride_dst);
    GF_CHECK_LAUNCH(fn);
    return GF_OK;
}

extern "C" GF_API int gf_transpose_pad(const void* src, int64_t ld_src, void* dst, int64_t R, int64_t C, int64_t rpad,
                                       void* stream) {
    GF_CHECK_ARG(src && dst && R > 0 && C > 0 && rpad >= R && ld_src >= C, "gf_transpose_pad: bad arguments");
    return transpose_pad_launch("gf_transpose_pad", src, ld_src, 0, dst, 0, R, C, rpad, 1, stream);
}

// `batch` matrices in one launch: matrix b is src + b stride_src ([R, C] with row pitch ld_src: heads side by side in one [R, H C] tensor
// have stride_src = C) -> dst + b stride_dst ([C, rpad]); element strides.
extern "C" GF_API int gf_transpose_pad_batched(const void* src, int64_t ld_src, int64_t stride_src, void* dst, int64_t stride_dst, int64_t R,
                                               int64_t C, int64_t rpad, int64_t batch, void* stream) {
    GF_CHECK_ARG(src && dst && R > 0 && C > 0 && rpad >= R && ld_src >= C && batch > 0 && batch < 65536 && stride_src >= 0 &&
                     stride_dst >= C * rpad, "gf_transpose_pad_batched: bad arguments");
    return transpose_pad_launch("gf_transpose_pad_batched", src, ld_src, stride_src, dst, stride_dst, R, C, rpad, batch, stream);
}

extern "C" GF_API int gf_vae_tile_blend(void* values, void* weight, const void* tile, int64_t nch, int64_t T, int64_t th,
                                        int64_t tw, int64_t tc, int64_t H, int64_t W, int64_t y0, int64_t x0, int top,
                                        int bottom, int left, int right, int64_t border_h, int64_t border_w,
                                        void* stream) {
    GF_CHECK_ARG(values && weight && tile && T > 0 && th > 0 && tw > 0 && nch > 0 && tc >= nch,
                 "gf_vae_tile_blend: bad arguments");
    GF_CHECK_ARG(y0 >= 0 && x0 >= 0 && y0 + th <= H && x0 + tw <= W, "gf_vae_tile_blend: tile outside the frame");
    GF_CHECK_ARG(border_h > 0 && border_w > 0, "gf_vae_tile_blend: borders must be positive");
    hipLaunchKernelGGL(tile_blend_kernel, dim3(vgrid(T * th * tw)), dim3(VT), 0, (hipStream_t)stream, (u16*)values,
                       (u16*)weight, (const u16*)tile, (int)nch, (int)T, (int)th, (int)tw, (int)tc, (int)H, (int)W,
                       (int)y0, (int)x0, top, bottom, left, right, (int)border_h, (int)border_w);
    GF_CHECK_LAUNCH("gf_vae_tile_blend");
    return GF_OK;
}

extern "C" GF_API int gf_vae_tile_finalize(void* values, const void* weight, int64_t planes, int64_t hw, int clamp1,
                                           void* stream) {
    GF_CHECK_ARG(values && weight && planes > 0 && hw > 0, "gf_vae_tile_finalize: bad arguments");
    hipLaunchKernelGGL(tile_finalize_kernel, dim3(vgrid(planes * hw)), dim3(VT), 0, (hipStream_t)stream, (u16*)values,
                       (const u16*)weight, (long)planes, (long)hw, clamp1 ? 1 : 0);
    GF_CHECK_LAUNCH("gf_vae_tile_finalize");
    return GF_OK;
}
