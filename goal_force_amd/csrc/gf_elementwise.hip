// gf_elementwise.hip — small HBM-bound kernels around the DiT forward and the sampler:
// CFG + Euler step, activations, residual add, (1,2,2) patch gather / scatter, force maps.
// All are grid-stride, 16 B per lane where the layout allows it.
#include "gf_common.h"

namespace {

constexpr int EW_THREADS = 256;
static inline unsigned ew_grid(long work_items) {
    long b = (work_items + EW_THREADS - 1) / EW_THREADS;
    if (b < 1) b = 1;
    if (b > 256 * 8) b = 256 * 8;  // 8 blocks per CU, grid-stride the rest
    return (unsigned)b;
}

// latents <- latents + (nega + cfg*(posi-nega)) * dsigma, every op rounded to bf16 like the
// reference's eager bf16 arithmetic (GF:716, FM:81).
__global__ __launch_bounds__(EW_THREADS) void cfg_euler_kernel(u16* __restrict__ lat, const u16* __restrict__ posi,
                                                               const u16* __restrict__ nega, float cfg, float dsigma,
                                                               long n) {
    const long nvec = n >> 3;
    const long stride = (long)gridDim.x * EW_THREADS;
    for (long i = (long)blockIdx.x * EW_THREADS + threadIdx.x; i < nvec; i += stride) {
        u16x8 l = reinterpret_cast<const u16x8*>(lat)[i];
        const u16x8 p = reinterpret_cast<const u16x8*>(posi)[i];
        u16x8 q;
        if (nega) q = reinterpret_cast<const u16x8*>(nega)[i];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float pred = bf2f(p[j]);
            if (nega) {
                const float ng = bf2f(q[j]);
                pred = rbf(ng + rbf(cfg * rbf(pred - ng)));
            }
            l[j] = f2bf(bf2f(l[j]) + rbf(pred * dsigma));
        }
        reinterpret_cast<u16x8*>(lat)[i] = l;
    }
    // tail (n % 8 elements)
    if (blockIdx.x == 0 && threadIdx.x < (n & 7)) {
        const long i = (nvec << 3) + threadIdx.x;
        float pred = bf2f(posi[i]);
        if (nega) {
            const float ng = bf2f(nega[i]);
            pred = rbf(ng + rbf(cfg * rbf(pred - ng)));
        }
        lat[i] = f2bf(bf2f(lat[i]) + rbf(pred * dsigma));
    }
}

template <int KIND>
__global__ __launch_bounds__(EW_THREADS) void act_kernel(const u16* __restrict__ x, u16* __restrict__ out, long n) {
    const long stride = (long)gridDim.x * EW_THREADS;
    for (long i = (long)blockIdx.x * EW_THREADS + threadIdx.x; i < n; i += stride) {
        const float v = bf2f(x[i]);
        out[i] = f2bf(KIND == 0 ? v / (1.0f + expf(-v)) : gelu_tanh_f(v));
    }
}

__global__ __launch_bounds__(EW_THREADS) void add_kernel(const u16* __restrict__ a, const u16* __restrict__ b,
                                                         u16* __restrict__ out, long n) {
    const long nvec = n >> 3;
    const long stride = (long)gridDim.x * EW_THREADS;
    for (long i = (long)blockIdx.x * EW_THREADS + threadIdx.x; i < nvec; i += stride) {
        const u16x8 x = reinterpret_cast<const u16x8*>(a)[i];
        const u16x8 y = reinterpret_cast<const u16x8*>(b)[i];
        u16x8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = f2bf(bf2f(x[j]) + bf2f(y[j]));
        reinterpret_cast<u16x8*>(out)[i] = o;
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 7)) {
        const long i = (nvec << 3) + threadIdx.x;
        out[i] = f2bf(bf2f(a[i]) + bf2f(b[i]));
    }
}

// modulate (DIT:64-65) as the reference's eager bf16 graph: t = bf16(1 + scale); y = bf16(x * t); out = bf16(y + shift);
// scale / shift are [dim] vectors (the reference's [B, 1, D] operands at B = 1).  dim % 8 == 0.
__global__ __launch_bounds__(EW_THREADS) void modulate_kernel(const u16* __restrict__ x, u16* __restrict__ out,
                                                              const u16* __restrict__ scale, const u16* __restrict__ shift,
                                                              long rows, int dim, long x_stride, long out_stride) {
    const int cpr = dim >> 3;
    const long total = rows * cpr;
    const long stride = (long)gridDim.x * EW_THREADS;
    for (long i = (long)blockIdx.x * EW_THREADS + threadIdx.x; i < total; i += stride) {
        const long r = i / cpr;
        const int c = (int)(i - r * cpr) << 3;
        const u16x8 xv = *reinterpret_cast<const u16x8*>(x + r * x_stride + c);
        const u16x8 sc = *reinterpret_cast<const u16x8*>(scale + c);
        const u16x8 sh = *reinterpret_cast<const u16x8*>(shift + c);
        u16x8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = f2bf(rbf(bf2f(xv[j]) * rbf(1.0f + bf2f(sc[j]))) + bf2f(sh[j]));
        *reinterpret_cast<u16x8*>(out + r * out_stride + c) = o;
    }
}

// GateModule.forward (DIT:189-194): x + gate * residual in the reference's eager bf16 order — bf16(gate * residual) first, then the
// add rounds again.  gate [dim] is the block's modulation row (batch 1).  The hot path fuses this into the GEMM epilogue
// (GF_EPI_BIAS_GATE_RESID); this kernel backs the module's own forward (B3).
__global__ __launch_bounds__(EW_THREADS) void gate_residual_kernel(const u16* __restrict__ x, const u16* __restrict__ gate,
                                                                   const u16* __restrict__ residual, u16* __restrict__ out, long rows,
                                                                   int dim, long x_stride, long r_stride, long out_stride) {
    const int cpr = dim >> 3;
    const long total = rows * cpr;
    const long stride = (long)gridDim.x * EW_THREADS;
    for (long i = (long)blockIdx.x * EW_THREADS + threadIdx.x; i < total; i += stride) {
        const long r = i / cpr;
        const int c = (int)(i - r * cpr) << 3;
        const u16x8 xv = *reinterpret_cast<const u16x8*>(x + r * x_stride + c);
        const u16x8 rv = *reinterpret_cast<const u16x8*>(residual + r * r_stride + c);
        const u16x8 gv = *reinterpret_cast<const u16x8*>(gate + c);
        u16x8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = f2bf(bf2f(xv[j]) + rbf(bf2f(gv[j]) * bf2f(rv[j])));
        *reinterpret_cast<u16x8*>(out + r * out_stride + c) = o;
    }
}

// rope_apply (DIT:92-97) alone: adjacent pairs of every head rotated by the token's phases (fp32 cos / sin tables [rows, head_dim/2];
// the reference multiplies in complex128 and rounds once to bf16).  The fused form on the hot path is gf_rmsnorm_rope.
__global__ __launch_bounds__(EW_THREADS) void rope_apply_kernel(const u16* __restrict__ x, u16* __restrict__ out,
                                                                const float* __restrict__ cos_tab, const float* __restrict__ sin_tab,
                                                                long rows, int dim, int head_dim, long x_stride, long out_stride) {
    const int cpr = dim >> 3, half = head_dim >> 1;
    const long total = rows * cpr;
    const long stride = (long)gridDim.x * EW_THREADS;
    for (long i = (long)blockIdx.x * EW_THREADS + threadIdx.x; i < total; i += stride) {
        const long r = i / cpr;
        const int c = (int)(i - r * cpr) << 3;
        const u16x8 xv = *reinterpret_cast<const u16x8*>(x + r * x_stride + c);
        const int p0 = (c % head_dim) >> 1;
        const f32x4 cs = *reinterpret_cast<const f32x4*>(cos_tab + r * half + p0);
        const f32x4 sn = *reinterpret_cast<const f32x4*>(sin_tab + r * half + p0);
        u16x8 o;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const float a = bf2f(xv[2 * p]), b = bf2f(xv[2 * p + 1]);
            o[2 * p] = f2bf(__builtin_fmaf(a, cs[p], -(b * sn[p])));
            o[2 * p + 1] = f2bf(__builtin_fmaf(a, sn[p], b * cs[p]));
        }
        *reinterpret_cast<u16x8*>(out + r * out_stride + c) = o;
    }
}

// out[i, :] = bf16(param[i, :] + t[i % t_rows, :]); rows whose bit is set in onep_mask become bf16(1 + .)
__global__ __launch_bounds__(EW_THREADS) void modulation_kernel(const u16* __restrict__ param,
                                                                const u16* __restrict__ t, u16* __restrict__ out,
                                                                int k, int dim, int t_rows, unsigned onep_mask) {
    const long total = (long)k * dim;
    const long stride = (long)gridDim.x * EW_THREADS;
    for (long i = (long)blockIdx.x * EW_THREADS + threadIdx.x; i < total; i += stride) {
        const int row = (int)(i / dim), col = (int)(i % dim);
        float v = rbf(bf2f(param[i]) + bf2f(t[(long)(row % t_rows) * dim + col]));
        if ((onep_mask >> row) & 1u) v = rbf(1.0f + v);
        out[i] = f2bf(v);
    }
}

// out[token, c*4 + dy*2 + dx] = src[c][f][2hh+dy][2ww+dx]; one thread per (token, channel slot),
// channel fastest so the 8-byte stores of a token row are contiguous.
__global__ __launch_bounds__(EW_THREADS) void patchify_kernel(const u16* __restrict__ s0, int c0,
                                                              const u16* __restrict__ s1, int c1,
                                                              u16* __restrict__ out, int F, int H, int W, int kpad) {
    const int h2 = H >> 1, w2 = W >> 1;
    const int slots = kpad >> 2;
    const long total = (long)F * h2 * w2 * slots;
    const long plane = (long)H * W, vol = (long)F * plane;
    const long stride = (long)gridDim.x * EW_THREADS;
    for (long i = (long)blockIdx.x * EW_THREADS + threadIdx.x; i < total; i += stride) {
        const int c = (int)(i % slots);
        const long tok = i / slots;
        u16x4 o = {0, 0, 0, 0};
        if (c < c0 + c1) {
            const int ww = (int)(tok % w2);
            const int hh = (int)((tok / w2) % h2);
            const int f = (int)(tok / ((long)w2 * h2));
            const u16* src = (c < c0) ? (s0 + (long)c * vol) : (s1 + (long)(c - c0) * vol);
            const u16* p = src + (long)f * plane + (long)(2 * hh) * W + 2 * ww;
            o[0] = p[0];
            o[1] = p[1];
            o[2] = p[W];
            o[3] = p[W + 1];
        }
        *reinterpret_cast<u16x4*>(out + tok * kpad + (c << 2)) = o;
    }
}

// tokens[(f,h,w), (y*2+z)*c + ch] -> out[ch][f][2h+y][2w+z]; one thread per output pair (z=0,1).
__global__ __launch_bounds__(EW_THREADS) void unpatchify_kernel(const u16* __restrict__ tok, u16* __restrict__ out,
                                                                int C, int f, int h, int w) {
    const int H = 2 * h, W = 2 * w;
    const long total = (long)C * f * H * w;
    const long stride = (long)gridDim.x * EW_THREADS;
    for (long i = (long)blockIdx.x * EW_THREADS + threadIdx.x; i < total; i += stride) {
        const int ww = (int)(i % w);
        long r = i / w;
        const int yy = (int)(r % H);
        r /= H;
        const int ff = (int)(r % f);
        const int ch = (int)(r / f);
        const int hh = yy >> 1, y = yy & 1;
        const long t = ((long)ff * h + hh) * w + ww;
        const u16* tp = tok + t * (4L * C) + (long)(y * 2) * C + ch;
        const unsigned v = (unsigned)tp[0] | ((unsigned)tp[C] << 16);
        *reinterpret_cast<unsigned*>(out + (((long)ch * f + ff) * H + yy) * W + 2 * ww) = v;
    }
}

// Control-signal video, THWC bf16.  One thread per pixel, all blobs summed in order in fp32.
__global__ __launch_bounds__(EW_THREADS) void force_map_kernel(u16* __restrict__ out, int frames, int H, int W,
                                                               const int* __restrict__ channels,
                                                               const float* __restrict__ params,
                                                               const float* __restrict__ centers, int n_blobs,
                                                               int clamp01) {
    const long total = (long)frames * H * W;
    const long stride = (long)gridDim.x * EW_THREADS;
    for (long i = (long)blockIdx.x * EW_THREADS + threadIdx.x; i < total; i += stride) {
        const int xg = (int)(i % W);
        const int yg = (int)((i / W) % H);
        const int t = (int)(i / ((long)W * H));
        float acc[3] = {0.f, 0.f, 0.f};
        for (int b = 0; b < n_blobs; ++b) {
            const float cx = centers[((long)b * frames + t) * 2 + 0];
            const float cy = centers[((long)b * frames + t) * 2 + 1];
            const float dx = (float)xg - cx, dy = (float)yg - cy;
            const float sq = dx * dx + dy * dy;
            const float g = params[2 * b + 1] * expf(-sq / params[2 * b]);
            const int ch = channels[b];
            if (ch == 0) acc[0] += g;
            else if (ch == 1) acc[1] += g;
            else acc[2] += g;
        }
        u16* o = out + i * 3;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float v = acc[c];
            if (clamp01) v = fminf(fmaxf(v, 0.f), 1.f);
            o[c] = f2bf(v);
        }
    }
}

}  // namespace

extern "C" GF_API int gf_cfg_euler_step(void* latents, const void* posi, const void* nega, float cfg_scale, float dsigma,
                                 int64_t n, void* stream) {
    GF_CHECK_ARG(latents && posi && n >= 0, "gf_cfg_euler_step: null pointer or negative n");
    GF_CHECK_ARG(gf_aligned16(latents) && gf_aligned16(posi) && (!nega || gf_aligned16(nega)),
                 "gf_cfg_euler_step: 16-byte alignment required");
    if (n == 0) return GF_OK;
    hipLaunchKernelGGL(cfg_euler_kernel, dim3(ew_grid(n >> 3)), dim3(EW_THREADS), 0, (hipStream_t)stream,
                       (u16*)latents, (const u16*)posi, (const u16*)nega, cfg_scale, dsigma, (long)n);
    GF_CHECK_LAUNCH("gf_cfg_euler_step");
    return GF_OK;
}

extern "C" GF_API int gf_act(const void* x, void* out, int64_t n, int kind, void* stream) {
    GF_CHECK_ARG(x && out && n >= 0, "gf_act: null pointer or negative n");
    GF_CHECK_ARG(kind == 0 || kind == 1, "gf_act: kind must be 0 (silu) or 1 (gelu_tanh), got %d", kind);
    if (n == 0) return GF_OK;
    if (kind == 0)
        hipLaunchKernelGGL(act_kernel<0>, dim3(ew_grid(n)), dim3(EW_THREADS), 0, (hipStream_t)stream,
                           (const u16*)x, (u16*)out, (long)n);
    else
        hipLaunchKernelGGL(act_kernel<1>, dim3(ew_grid(n)), dim3(EW_THREADS), 0, (hipStream_t)stream,
                           (const u16*)x, (u16*)out, (long)n);
    GF_CHECK_LAUNCH("gf_act");
    return GF_OK;
}

extern "C" GF_API int gf_add_bf16(const void* a, const void* b, void* out, int64_t n, void* stream) {
    GF_CHECK_ARG(a && b && out && n >= 0, "gf_add_bf16: null pointer or negative n");
    GF_CHECK_ARG(gf_aligned16(a) && gf_aligned16(b) && gf_aligned16(out), "gf_add_bf16: 16-byte alignment required");
    if (n == 0) return GF_OK;
    hipLaunchKernelGGL(add_kernel, dim3(ew_grid(n >> 3)), dim3(EW_THREADS), 0, (hipStream_t)stream, (const u16*)a,
                       (const u16*)b, (u16*)out, (long)n);
    GF_CHECK_LAUNCH("gf_add_bf16");
    return GF_OK;
}

extern "C" GF_API int gf_modulate(const void* x, void* out, const void* scale, const void* shift, int64_t rows, int64_t dim,
                                  int64_t x_stride, int64_t out_stride, void* stream) {
    GF_CHECK_ARG(x && out && scale && shift && rows >= 0, "gf_modulate: null pointer");
    GF_CHECK_ARG(dim > 0 && dim % 8 == 0 && x_stride % 8 == 0 && out_stride % 8 == 0 && gf_aligned16(x) && gf_aligned16(out) &&
                     gf_aligned16(scale) && gf_aligned16(shift), "gf_modulate: dim %% 8 == 0 and 16-byte aligned rows / vectors required");
    if (rows == 0) return GF_OK;
    hipLaunchKernelGGL(modulate_kernel, dim3(ew_grid(rows * (dim >> 3))), dim3(EW_THREADS), 0, (hipStream_t)stream, (const u16*)x,
                       (u16*)out, (const u16*)scale, (const u16*)shift, (long)rows, (int)dim, (long)x_stride, (long)out_stride);
    GF_CHECK_LAUNCH("gf_modulate");
    return GF_OK;
}

extern "C" GF_API int gf_gate_residual(const void* x, const void* gate, const void* residual, void* out, int64_t rows, int64_t dim,
                                       int64_t x_stride, int64_t r_stride, int64_t out_stride, void* stream) {
    GF_CHECK_ARG(x && gate && residual && out && rows >= 0, "gf_gate_residual: null pointer");
    GF_CHECK_ARG(dim > 0 && dim % 8 == 0 && x_stride % 8 == 0 && r_stride % 8 == 0 && out_stride % 8 == 0 && gf_aligned16(x) &&
                     gf_aligned16(gate) && gf_aligned16(residual) && gf_aligned16(out),
                 "gf_gate_residual: dim %% 8 == 0 and 16-byte aligned rows / vectors required");
    if (rows == 0) return GF_OK;
    hipLaunchKernelGGL(gate_residual_kernel, dim3(ew_grid(rows * (dim >> 3))), dim3(EW_THREADS), 0, (hipStream_t)stream, (const u16*)x,
                       (const u16*)gate, (const u16*)residual, (u16*)out, (long)rows, (int)dim, (long)x_stride, (long)r_stride,
                       (long)out_stride);
    GF_CHECK_LAUNCH("gf_gate_residual");
    return GF_OK;
}

extern "C" GF_API int gf_rope_apply(const void* x, void* out, const float* cos_tab, const float* sin_tab, int64_t rows,
                                    int64_t dim, int64_t head_dim, int64_t x_stride, int64_t out_stride, void* stream) {
    GF_CHECK_ARG(x && out && cos_tab && sin_tab && rows >= 0, "gf_rope_apply: null pointer");
    GF_CHECK_ARG(dim > 0 && head_dim > 0 && head_dim % 8 == 0 && dim % head_dim == 0, "gf_rope_apply: head_dim must be a multiple of 8 dividing dim");
    GF_CHECK_ARG(x_stride % 8 == 0 && out_stride % 8 == 0 && gf_aligned16(x) && gf_aligned16(out) && gf_aligned16(cos_tab) &&
                     gf_aligned16(sin_tab), "gf_rope_apply: 16-byte alignment required");
    if (rows == 0) return GF_OK;
    hipLaunchKernelGGL(rope_apply_kernel, dim3(ew_grid(rows * (dim >> 3))), dim3(EW_THREADS), 0, (hipStream_t)stream, (const u16*)x,
                       (u16*)out, cos_tab, sin_tab, (long)rows, (int)dim, (int)head_dim, (long)x_stride, (long)out_stride);
    GF_CHECK_LAUNCH("gf_rope_apply");
    return GF_OK;
}

extern "C" GF_API int gf_modulation(const void* param, const void* t, void* out, int64_t k, int64_t dim,
                                    int64_t t_rows, uint32_t onep_mask, void* stream) {
    GF_CHECK_ARG(param && t && out, "gf_modulation: null pointer");
    GF_CHECK_ARG(k > 0 && k <= 32 && dim > 0 && t_rows > 0 && t_rows <= k, "gf_modulation: bad k=%ld dim=%ld t_rows=%ld",
                 (long)k, (long)dim, (long)t_rows);
    hipLaunchKernelGGL(modulation_kernel, dim3(ew_grid(k * dim)), dim3(EW_THREADS), 0, (hipStream_t)stream,
                       (const u16*)param, (const u16*)t, (u16*)out, (int)k, (int)dim, (int)t_rows, onep_mask);
    GF_CHECK_LAUNCH("gf_modulation");
    return GF_OK;
}

extern "C" GF_API int gf_patchify_im2col(const void* src0, int64_t c0, const void* src1, int64_t c1, void* out, int64_t F,
                                  int64_t H, int64_t W, int64_t kpad, void* stream) {
    GF_CHECK_ARG(src0 && out && c0 > 0 && c1 >= 0 && (c1 == 0 || src1), "gf_patchify_im2col: bad sources");
    GF_CHECK_ARG(F > 0 && H > 0 && W > 0 && H % 2 == 0 && W % 2 == 0, "gf_patchify_im2col: H and W must be even");
    GF_CHECK_ARG(kpad % 8 == 0 && kpad >= (c0 + c1) * 4, "gf_patchify_im2col: kpad=%ld too small or not a multiple of 8",
                 (long)kpad);
    const long total = F * (H / 2) * (W / 2) * (kpad / 4);
    hipLaunchKernelGGL(patchify_kernel, dim3(ew_grid(total)), dim3(EW_THREADS), 0, (hipStream_t)stream,
                       (const u16*)src0, (int)c0, (const u16*)src1, (int)c1, (u16*)out, (int)F, (int)H, (int)W,
                       (int)kpad);
    GF_CHECK_LAUNCH("gf_patchify_im2col");
    return GF_OK;
}

extern "C" GF_API int gf_unpatchify(const void* tokens, void* out, int64_t c, int64_t f, int64_t h, int64_t w, void* stream) {
    GF_CHECK_ARG(tokens && out && c > 0 && f > 0 && h > 0 && w > 0, "gf_unpatchify: bad arguments");
    GF_CHECK_ARG((((uintptr_t)out) & 3u) == 0, "gf_unpatchify: out must be 4-byte aligned");
    const long total = c * f * 2 * h * w;
    hipLaunchKernelGGL(unpatchify_kernel, dim3(ew_grid(total)), dim3(EW_THREADS), 0, (hipStream_t)stream,
                       (const u16*)tokens, (u16*)out, (int)c, (int)f, (int)h, (int)w);
    GF_CHECK_LAUNCH("gf_unpatchify");
    return GF_OK;
}

extern "C" GF_API int gf_force_map(void* out, int64_t frames, int64_t H, int64_t W, const int32_t* channels,
                            const float* params, const float* centers, int64_t n_blobs, int clamp01, void* stream) {
    GF_CHECK_ARG(out && frames > 0 && H > 0 && W > 0 && n_blobs >= 0, "gf_force_map: bad arguments");
    GF_CHECK_ARG(n_blobs == 0 || (channels && params && centers), "gf_force_map: null blob arrays");
    hipLaunchKernelGGL(force_map_kernel, dim3(ew_grid(frames * H * W)), dim3(EW_THREADS), 0, (hipStream_t)stream,
                       (u16*)out, (int)frames, (int)H, (int)W, (const int*)channels, params, centers, (int)n_blobs,
                       clamp01);
    GF_CHECK_LAUNCH("gf_force_map");
    return GF_OK;
}
