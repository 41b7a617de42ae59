// gf_backward.hip — backward of the HBM-bound row / elementwise ops of a DiT block, plus the loss and the optimiser
// step of the ControlNet training step (reference: training_loss src/goal_force/wan_video_new.py:180-193,
// launch_training_task src/goal_force/utils.py:734-826: MSE loss, AdamW, bf16 parameters).
// Gradients are formed in fp32 from the bf16 forward operands and rounded once to bf16 (what torch autograd hands back
// for a bf16 graph); per-column parameter gradients (modulation rows, norm weights, biases, gates) are accumulated over
// the token rows in fp32 with one atomic add per column per workgroup into caller-zeroed [dim] buffers.
//
//   LayerNorm(+affine | +modulate)   y = rbf(xhat [*w + b]) [* scale1p + shift]     DIT:206-208, 64-65; VRAM:78-92
//   RMSNorm(+RoPE)                   y = rope(rbf(rbf(x*rinv) * w))                 DIT:100-111, 92-97
//   GELU-tanh, gated residual        f = gelu(u);  out = resid + rbf(gate * y)      DIT:209-210, 189-194, 226-229
#include "gf_common.h"

namespace {

constexpr int ROW_THREADS = 128;
constexpr int ROW_NCH = 8;      // 16-byte chunks per thread -> dim <= 8192
constexpr int ROWS_PER_WG = 16; // rows one workgroup walks: its column partial sums stay in registers

// dx of LayerNorm over the row; g = per-column multiplier applied to xhat (affine weight and/or 1+scale), may be null.
//   affine  (weight != null): y = xhat*w + b          -> dxhat = dy*w;         dw += dy*xhat,  db += dy
//   modulate (scale1p != null): y = xhat*s1p + shift  -> dxhat = dy*s1p;       ds1p += dy*xhat, dshift += dy
// (the block never uses both on one LayerNorm).  dg_acc / db_acc are fp32 [dim] or null.
__global__ __launch_bounds__(ROW_THREADS) void layernorm_bwd_kernel(const u16* __restrict__ x, const u16* __restrict__ dy,
                                                                    const u16* __restrict__ g, u16* __restrict__ dx,
                                                                    float* __restrict__ dg_acc, float* __restrict__ db_acc,
                                                                    int rows, int dim, long x_stride, long dy_stride,
                                                                    long dx_stride, float eps) {
    __shared__ float red[ROW_THREADS / 64];
    const int tid = threadIdx.x;
    const int nchunks = dim >> 3;
    float gsum[ROW_NCH][8], bsum[ROW_NCH][8];
#pragma unroll
    for (int i = 0; i < ROW_NCH; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) gsum[i][j] = bsum[i][j] = 0.f;
    const long row0 = (long)blockIdx.x * ROWS_PER_WG;
    for (int rr = 0; rr < ROWS_PER_WG && row0 + rr < rows; ++rr) {
        const long row = row0 + rr;
        const u16* xr = x + row * x_stride;
        const u16* dr = dy + row * dy_stride;
        u16x8 v[ROW_NCH], d[ROW_NCH];
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < ROW_NCH; ++i) {
            const int c = tid + i * ROW_THREADS;
            if (c < nchunks) {
                v[i] = *reinterpret_cast<const u16x8*>(xr + (c << 3));
                d[i] = *reinterpret_cast<const u16x8*>(dr + (c << 3));
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
                    const float t = bf2f(v[i][j]) - mean;
                    q += t * t;
                }
            }
        }
        const float rstd = 1.0f / sqrtf(block_sum<ROW_THREADS>(q, red) / (float)dim + eps);
        // m1 = mean(dxhat), m2 = mean(dxhat * xhat)
        float a1 = 0.f, a2 = 0.f;
#pragma unroll
        for (int i = 0; i < ROW_NCH; ++i) {
            const int c = tid + i * ROW_THREADS;
            if (c < nchunks) {
                u16x8 g8;
                if (g) g8 = *reinterpret_cast<const u16x8*>(g + (c << 3));
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float xh = (bf2f(v[i][j]) - mean) * rstd;
                    const float dyv = bf2f(d[i][j]);
                    const float dxh = g ? dyv * bf2f(g8[j]) : dyv;
                    a1 += dxh;
                    a2 += dxh * xh;
                    gsum[i][j] += dyv * xh;
                    bsum[i][j] += dyv;
                }
            }
        }
        const float m1 = block_sum<ROW_THREADS>(a1, red) / (float)dim;
        const float m2 = block_sum<ROW_THREADS>(a2, red) / (float)dim;
        u16* orow = dx + row * dx_stride;
#pragma unroll
        for (int i = 0; i < ROW_NCH; ++i) {
            const int c = tid + i * ROW_THREADS;
            if (c < nchunks) {
                u16x8 g8, o;
                if (g) g8 = *reinterpret_cast<const u16x8*>(g + (c << 3));
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float xh = (bf2f(v[i][j]) - mean) * rstd;
                    const float dxh = g ? bf2f(d[i][j]) * bf2f(g8[j]) : bf2f(d[i][j]);
                    o[j] = f2bf(rstd * (dxh - m1 - xh * m2));
                }
                *reinterpret_cast<u16x8*>(orow + (c << 3)) = o;
            }
        }
    }
    if (dg_acc || db_acc) {
#pragma unroll
        for (int i = 0; i < ROW_NCH; ++i) {
            const int c = tid + i * ROW_THREADS;
            if (c < nchunks) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    if (dg_acc) atomicAdd(dg_acc + (c << 3) + j, gsum[i][j]);
                    if (db_acc) atomicAdd(db_acc + (c << 3) + j, bsum[i][j]);
                }
            }
        }
    }
}

// Wave-per-row forms of the two row backward kernels (round 3): a row of up to 5120 elements sits in the registers of ONE wave
// (10 chunks of 16 bytes per lane), the four row sums are wave shuffles — no LDS, no barrier.  The block-per-16-rows kernels above /
// below take 8 barriers per row at two waves per workgroup and ran at a quarter of the HBM rate (0.69 / 0.53 ms per [32760, 5120]
// call against ~0.17 for the bytes).  They serve the calls that want NO column sums (a frozen block: 4 of 5 calls of a training step):
// 0.37 ms.  With column sums the block kernels stay: keeping 160 partial sums per lane in registers (1.3 ms) or adding them into LDS
// with ds_add_f32 (1.9 ms) both lost to them.  (A frozen block's dx therefore differs from a trainable one's in the last bit of some
// elements: the row sums are added in another order.)
constexpr int WROW_NCH = 10, WROW_WAVES = 4, WROW_DIM = WROW_NCH * 64 * 8;
constexpr int WROW_ROWS = 2;      // rows per wave
__global__ __launch_bounds__(64 * WROW_WAVES) void layernorm_bwd_wave_kernel(const u16* __restrict__ x, const u16* __restrict__ dy,
                                                                             const u16* __restrict__ g, u16* __restrict__ dx,
                                                                             int rows, int dim, long x_stride, long dy_stride,
                                                                             long dx_stride, float eps) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nchunks = dim >> 3;
    u16x8 g8[WROW_NCH];
#pragma unroll
    for (int i = 0; i < WROW_NCH; ++i) {
        const int c = lane + 64 * i;
        g8[i] = (g && c < nchunks) ? *reinterpret_cast<const u16x8*>(g + (c << 3)) : u16x8{0, 0, 0, 0, 0, 0, 0, 0};
    }
    const long row0 = ((long)blockIdx.x * WROW_WAVES + wave) * WROW_ROWS;
    const float inv_dim = 1.0f / (float)dim;
    for (int rr = 0; rr < WROW_ROWS && row0 + rr < rows; ++rr) {
        const long row = row0 + rr;
        const u16* xr = x + row * x_stride;
        const u16* dr = dy + row * dy_stride;
        u16x8 v[WROW_NCH], d[WROW_NCH];
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < WROW_NCH; ++i) {
            const int c = lane + 64 * i;
            if (c < nchunks) {
                v[i] = *reinterpret_cast<const u16x8*>(xr + (c << 3));
                d[i] = *reinterpret_cast<const u16x8*>(dr + (c << 3));
#pragma unroll
                for (int j = 0; j < 8; ++j) s += bf2f(v[i][j]);
            }
        }
        const float mean = wave_sum(s) * inv_dim;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < WROW_NCH; ++i)
            if (lane + 64 * i < nchunks) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float t = bf2f(v[i][j]) - mean;
                    q += t * t;
                }
            }
        const float rstd = 1.0f / sqrtf(wave_sum(q) * inv_dim + eps);
        float a1 = 0.f, a2 = 0.f;      // m1 = mean(dxhat), m2 = mean(dxhat * xhat)
#pragma unroll
        for (int i = 0; i < WROW_NCH; ++i)
            if (lane + 64 * i < nchunks) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float xh = (bf2f(v[i][j]) - mean) * rstd;
                    const float dyv = bf2f(d[i][j]);
                    const float dxh = g ? dyv * bf2f(g8[i][j]) : dyv;
                    a1 += dxh;
                    a2 += dxh * xh;
                }
            }
        const float m1 = wave_sum(a1) * inv_dim, m2 = wave_sum(a2) * inv_dim;
        u16* orow = dx + row * dx_stride;
#pragma unroll
        for (int i = 0; i < WROW_NCH; ++i) {
            const int c = lane + 64 * i;
            if (c < nchunks) {
                u16x8 o;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float xh = (bf2f(v[i][j]) - mean) * rstd;
                    const float dxh = g ? bf2f(d[i][j]) * bf2f(g8[i][j]) : bf2f(d[i][j]);
                    o[j] = f2bf(rstd * (dxh - m1 - xh * m2));
                }
                *reinterpret_cast<u16x8*>(orow + (c << 3)) = o;
            }
        }
    }
}

__global__ __launch_bounds__(64 * WROW_WAVES) void rmsnorm_rope_bwd_wave_kernel(const u16* __restrict__ x, const u16* __restrict__ dy,
                                                                                const u16* __restrict__ weight,
                                                                                const float* __restrict__ cos_tab,
                                                                                const float* __restrict__ sin_tab, u16* __restrict__ dx,
                                                                                int rows, int dim, int head_dim, long x_stride,
                                                                                long dy_stride, long dx_stride, float eps) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nchunks = dim >> 3, half = head_dim >> 1;
    u16x8 w8[WROW_NCH];
#pragma unroll
    for (int i = 0; i < WROW_NCH; ++i) {
        const int c = lane + 64 * i;
        w8[i] = c < nchunks ? *reinterpret_cast<const u16x8*>(weight + (c << 3)) : u16x8{0, 0, 0, 0, 0, 0, 0, 0};
    }
    const long row0 = ((long)blockIdx.x * WROW_WAVES + wave) * WROW_ROWS;
    const float inv_dim = 1.0f / (float)dim;
    for (int rr = 0; rr < WROW_ROWS && row0 + rr < rows; ++rr) {
        const long row = row0 + rr;
        const u16* xr = x + row * x_stride;
        const u16* dr = dy + row * dy_stride;
        u16x8 v[WROW_NCH];
        float dz[WROW_NCH][8];
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < WROW_NCH; ++i) {
            const int c = lane + 64 * i;
            if (c < nchunks) {
                v[i] = *reinterpret_cast<const u16x8*>(xr + (c << 3));
                const u16x8 d8 = *reinterpret_cast<const u16x8*>(dr + (c << 3));
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float f = bf2f(v[i][j]);
                    s += f * f;
                    dz[i][j] = bf2f(d8[j]);
                }
                if (cos_tab) {
                    const int p0 = ((c << 3) % head_dim) >> 1;
                    const f32x4 cs = *reinterpret_cast<const f32x4*>(cos_tab + row * half + p0);
                    const f32x4 sn = *reinterpret_cast<const f32x4*>(sin_tab + row * half + p0);
#pragma unroll
                    for (int pp = 0; pp < 4; ++pp) {   // transpose of [[c,-s],[s,c]]
                        const float aa = dz[i][2 * pp], bb = dz[i][2 * pp + 1];
                        dz[i][2 * pp] = aa * cs[pp] + bb * sn[pp];
                        dz[i][2 * pp + 1] = -aa * sn[pp] + bb * cs[pp];
                    }
                }
            }
        }
        const float rstd = 1.0f / sqrtf(wave_sum(s) * inv_dim + eps);
        float a2 = 0.f;
#pragma unroll
        for (int i = 0; i < WROW_NCH; ++i)
            if (lane + 64 * i < nchunks) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float xn = bf2f(v[i][j]) * rstd;
                    dz[i][j] *= bf2f(w8[i][j]);          // dxn
                    a2 += dz[i][j] * xn;
                }
            }
        const float m2 = wave_sum(a2) * inv_dim;
        u16* orow = dx + row * dx_stride;
#pragma unroll
        for (int i = 0; i < WROW_NCH; ++i) {
            const int c = lane + 64 * i;
            if (c < nchunks) {
                u16x8 o;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float xn = bf2f(v[i][j]) * rstd;
                    o[j] = f2bf(rstd * (dz[i][j] - xn * m2));
                }
                *reinterpret_cast<u16x8*>(orow + (c << 3)) = o;
            }
        }
    }
}

// RMSNorm(+RoPE) backward.  x = the PRE-norm tensor, dy = gradient of the rotated output.
//   dz = rope^-1(dy) (rotation by -theta);  xn = x*rinv;  dw += dz*xn;  dxn = dz*w;  dx = rinv (dxn - xn mean(dxn xn))
__global__ __launch_bounds__(ROW_THREADS) void rmsnorm_rope_bwd_kernel(const u16* __restrict__ x, const u16* __restrict__ dy,
                                                                       const u16* __restrict__ weight,
                                                                       const float* __restrict__ cos_tab,
                                                                       const float* __restrict__ sin_tab, u16* __restrict__ dx,
                                                                       float* __restrict__ dw_acc, int rows, int dim,
                                                                       int head_dim, long x_stride, long dy_stride,
                                                                       long dx_stride, float eps) {
    __shared__ float red[ROW_THREADS / 64];
    const int tid = threadIdx.x;
    const int nchunks = dim >> 3;
    const int half = head_dim >> 1;
    float wsum[ROW_NCH][8];
#pragma unroll
    for (int i = 0; i < ROW_NCH; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) wsum[i][j] = 0.f;
    const long row0 = (long)blockIdx.x * ROWS_PER_WG;
    for (int rr = 0; rr < ROWS_PER_WG && row0 + rr < rows; ++rr) {
        const long row = row0 + rr;
        const u16* xr = x + row * x_stride;
        const u16* dr = dy + row * dy_stride;
        u16x8 v[ROW_NCH];
        float dz[ROW_NCH][8];
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < ROW_NCH; ++i) {
            const int c = tid + i * ROW_THREADS;
            if (c < nchunks) {
                v[i] = *reinterpret_cast<const u16x8*>(xr + (c << 3));
                const u16x8 d8 = *reinterpret_cast<const u16x8*>(dr + (c << 3));
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float f = bf2f(v[i][j]);
                    s += f * f;
                    dz[i][j] = bf2f(d8[j]);
                }
                if (cos_tab) {
                    const int p0 = ((c << 3) % head_dim) >> 1;
                    const f32x4 cs = *reinterpret_cast<const f32x4*>(cos_tab + row * half + p0);
                    const f32x4 sn = *reinterpret_cast<const f32x4*>(sin_tab + row * half + p0);
#pragma unroll
                    for (int p = 0; p < 4; ++p) {   // transpose of [[c,-s],[s,c]]
                        const float a = dz[i][2 * p], b = dz[i][2 * p + 1];
                        dz[i][2 * p] = a * cs[p] + b * sn[p];
                        dz[i][2 * p + 1] = -a * sn[p] + b * cs[p];
                    }
                }
            }
        }
        const float rstd = 1.0f / sqrtf(block_sum<ROW_THREADS>(s, red) / (float)dim + eps);
        float a2 = 0.f;
#pragma unroll
        for (int i = 0; i < ROW_NCH; ++i) {
            const int c = tid + i * ROW_THREADS;
            if (c < nchunks) {
                const u16x8 w8 = *reinterpret_cast<const u16x8*>(weight + (c << 3));
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float xn = bf2f(v[i][j]) * rstd;
                    wsum[i][j] += dz[i][j] * xn;
                    dz[i][j] *= bf2f(w8[j]);          // dxn
                    a2 += dz[i][j] * xn;
                }
            }
        }
        const float m2 = block_sum<ROW_THREADS>(a2, red) / (float)dim;
        u16* orow = dx + row * dx_stride;
#pragma unroll
        for (int i = 0; i < ROW_NCH; ++i) {
            const int c = tid + i * ROW_THREADS;
            if (c < nchunks) {
                u16x8 o;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float xn = bf2f(v[i][j]) * rstd;
                    o[j] = f2bf(rstd * (dz[i][j] - xn * m2));
                }
                *reinterpret_cast<u16x8*>(orow + (c << 3)) = o;
            }
        }
    }
    if (dw_acc) {
#pragma unroll
        for (int i = 0; i < ROW_NCH; ++i) {
            const int c = tid + i * ROW_THREADS;
            if (c < nchunks) {
#pragma unroll
                for (int j = 0; j < 8; ++j) atomicAdd(dw_acc + (c << 3) + j, wsum[i][j]);
            }
        }
    }
}

// acc[n] += sum over rows of a[r, n] * (b ? b[r, n] : 1);  optionally out[r, n] = bf16(a[r, n] * gate[n])
// (bias gradients; gate gradient + gated upstream gradient of out = resid + gate * y)
__global__ __launch_bounds__(256) void colsum_kernel(const u16* __restrict__ a, const u16* __restrict__ b,
                                                     const u16* __restrict__ gate, u16* __restrict__ out,
                                                     float* __restrict__ acc, int rows, int cols, long lda, long ldb, long ldo) {
    const int col = (blockIdx.x * 256 + threadIdx.x) * 8;
    if (col >= cols) return;
    const long row0 = (long)blockIdx.y * 64;
    float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    u16x8 g8;
    if (gate) g8 = *reinterpret_cast<const u16x8*>(gate + col);
    for (int rr = 0; rr < 64 && row0 + rr < rows; ++rr) {
        const long r = row0 + rr;
        const u16x8 a8 = *reinterpret_cast<const u16x8*>(a + r * lda + col);
        u16x8 b8, o;
        if (b) b8 = *reinterpret_cast<const u16x8*>(b + r * ldb + col);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float av = bf2f(a8[j]);
            s[j] += b ? av * bf2f(b8[j]) : av;
            if (out) o[j] = f2bf(av * bf2f(g8[j]));
        }
        if (out) *reinterpret_cast<u16x8*>(out + r * ldo + col) = o;
    }
    if (acc) {
#pragma unroll
        for (int j = 0; j < 8; ++j) atomicAdd(acc + col + j, s[j]);
    }
}

__device__ __forceinline__ float gelu_tanh_grad(float u) {
    const float k0 = 0.7978845608028654f, k1 = 0.044715f;
    const float t = tanhf(k0 * (u + k1 * u * u * u));
    return 0.5f * (1.0f + t) + 0.5f * u * (1.0f - t * t) * k0 * (1.0f + 3.0f * k1 * u * u);
}

// du = df * gelu_tanh'(u)   (kind 0)  |  du = df * silu'(u)  (kind 1)
__global__ __launch_bounds__(256) void act_bwd_kernel(const u16* __restrict__ u, const u16* __restrict__ df,
                                                      u16* __restrict__ du, long n, int kind) {
    const long i = ((long)blockIdx.x * 256 + threadIdx.x) * 8;
    if (i >= n) return;
    const u16x8 u8 = *reinterpret_cast<const u16x8*>(u + i);
    const u16x8 d8 = *reinterpret_cast<const u16x8*>(df + i);
    u16x8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float x = bf2f(u8[j]);
        float gr;
        if (kind == 0) {
            gr = gelu_tanh_grad(x);
        } else {
            const float sg = 1.0f / (1.0f + expf(-x));
            gr = sg * (1.0f + x * (1.0f - sg));
        }
        o[j] = f2bf(bf2f(d8[j]) * gr);
    }
    *reinterpret_cast<u16x8*>(du + i) = o;
}

// loss = weight * mean((pred - target)^2) (fp32 accumulate, F.mse_loss on .float()), dpred = weight * 2 (pred - target) / n
__global__ __launch_bounds__(256) void mse_kernel(const u16* __restrict__ pred, const u16* __restrict__ target,
                                                  u16* __restrict__ dpred, float* __restrict__ loss, long n, float weight) {
    __shared__ float red[4];
    float s = 0.f;
    const float k = 2.0f * weight / (float)n;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const float d = bf2f(pred[i]) - bf2f(target[i]);
        s += d * d;
        if (dpred) dpred[i] = f2bf(k * d);
    }
    const float tot = block_sum<256>(s, red);
    if (threadIdx.x == 0) atomicAdd(loss, tot * weight / (float)n);
}

// torch.optim.AdamW (decoupled weight decay), bf16 parameter, fp32 moments:
//   p *= 1 - lr*wd;  m = b1 m + (1-b1) g;  v = b2 v + (1-b2) g^2;  p -= lr/(1-b1^t) * m / (sqrt(v)/sqrt(1-b2^t) + eps)
__global__ __launch_bounds__(256) void adamw_kernel(u16* __restrict__ p, const u16* __restrict__ g, float* __restrict__ m,
                                                    float* __restrict__ v, long n, float lr, float b1, float b2, float eps,
                                                    float wd, float bc1, float bc2_sqrt, float grad_scale) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float gr = bf2f(g[i]) * grad_scale;
    float pv = bf2f(p[i]);
    pv = rbf(pv * (1.0f - lr * wd));                  // torch does p.mul_(1 - lr*wd) on the bf16 parameter
    const float mv = b1 * m[i] + (1.0f - b1) * gr;
    const float vv = b2 * v[i] + (1.0f - b2) * gr * gr;
    m[i] = mv;
    v[i] = vv;
    const float denom = sqrtf(vv) / bc2_sqrt + eps;
    p[i] = f2bf(pv - (lr / bc1) * (mv / denom));
}

// acc[0] += sum x^2  (global gradient norm of clip_grad_norm_, utils.py:806-808)
__global__ __launch_bounds__(256) void sumsq_kernel(const u16* __restrict__ x, float* __restrict__ acc, long n) {
    __shared__ float red[4];
    float s = 0.f;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const float v = bf2f(x[i]);
        s += v * v;
    }
    const float tot = block_sum<256>(s, red);
    if (threadIdx.x == 0) atomicAdd(acc, tot);
}

// dst = bf16(acc)  (column accumulators -> parameter gradients)
__global__ __launch_bounds__(256) void f32_to_bf16_kernel(const float* __restrict__ a, u16* __restrict__ o, long n) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) o[i] = f2bf(a[i]);
}

}  // namespace

extern "C" GF_API int gf_layernorm_bwd(const void* x, int64_t x_stride, const void* dy, int64_t dy_stride, const void* g,
                                       void* dx, int64_t dx_stride, float* dg_acc, float* db_acc, int64_t rows, int64_t dim,
                                       float eps, void* stream) {
    GF_CHECK_ARG(x && dy && dx, "gf_layernorm_bwd: null pointer");
    GF_CHECK_ARG(rows >= 0 && dim > 0 && dim % 8 == 0 && dim <= ROW_THREADS * ROW_NCH * 8, "gf_layernorm_bwd: dim=%ld unsupported",
                 (long)dim);
    GF_CHECK_ARG(x_stride % 8 == 0 && dy_stride % 8 == 0 && dx_stride % 8 == 0 && gf_aligned16(x) && gf_aligned16(dy) &&
                     gf_aligned16(dx) && (!g || gf_aligned16(g)),
                 "gf_layernorm_bwd: 16-byte alignment required");
    if (rows == 0) return GF_OK;
    if (dim <= WROW_DIM && !dg_acc && !db_acc) {      // no column sums wanted and a row fits one wave's registers
        const unsigned grid = (unsigned)((rows + WROW_WAVES * WROW_ROWS - 1) / (WROW_WAVES * WROW_ROWS));
        hipLaunchKernelGGL(layernorm_bwd_wave_kernel, dim3(grid), dim3(64 * WROW_WAVES), 0, (hipStream_t)stream, (const u16*)x, (const u16*)dy,
                           (const u16*)g, (u16*)dx, (int)rows, (int)dim, (long)x_stride, (long)dy_stride, (long)dx_stride, eps);
    } else
    hipLaunchKernelGGL(layernorm_bwd_kernel, dim3((unsigned)((rows + ROWS_PER_WG - 1) / ROWS_PER_WG)), dim3(ROW_THREADS), 0,
                       (hipStream_t)stream, (const u16*)x, (const u16*)dy, (const u16*)g, (u16*)dx, dg_acc, db_acc, (int)rows,
                       (int)dim, (long)x_stride, (long)dy_stride, (long)dx_stride, eps);
    GF_CHECK_LAUNCH("gf_layernorm_bwd");
    return GF_OK;
}

extern "C" GF_API int gf_rmsnorm_rope_bwd(const void* x, int64_t x_stride, const void* dy, int64_t dy_stride, const void* weight,
                                          const float* cos_tab, const float* sin_tab, void* dx, int64_t dx_stride,
                                          float* dw_acc, int64_t rows, int64_t dim, int64_t head_dim, float eps, void* stream) {
    GF_CHECK_ARG(x && dy && dx && weight, "gf_rmsnorm_rope_bwd: null pointer");
    GF_CHECK_ARG((cos_tab == nullptr) == (sin_tab == nullptr), "gf_rmsnorm_rope_bwd: cos and sin go together");
    GF_CHECK_ARG(rows >= 0 && dim > 0 && dim % 8 == 0 && dim <= ROW_THREADS * ROW_NCH * 8 && head_dim > 0 && head_dim % 8 == 0 &&
                     dim % head_dim == 0,
                 "gf_rmsnorm_rope_bwd: dim=%ld head_dim=%ld unsupported", (long)dim, (long)head_dim);
    GF_CHECK_ARG(x_stride % 8 == 0 && dy_stride % 8 == 0 && dx_stride % 8 == 0 && gf_aligned16(x) && gf_aligned16(dy) &&
                     gf_aligned16(dx) && gf_aligned16(weight) && (!cos_tab || (gf_aligned16(cos_tab) && gf_aligned16(sin_tab))),
                 "gf_rmsnorm_rope_bwd: 16-byte alignment required");
    if (rows == 0) return GF_OK;
    if (dim <= WROW_DIM && !dw_acc) {
        const unsigned grid = (unsigned)((rows + WROW_WAVES * WROW_ROWS - 1) / (WROW_WAVES * WROW_ROWS));
        hipLaunchKernelGGL(rmsnorm_rope_bwd_wave_kernel, dim3(grid), dim3(64 * WROW_WAVES), 0, (hipStream_t)stream, (const u16*)x, (const u16*)dy,
                           (const u16*)weight, cos_tab, sin_tab, (u16*)dx, (int)rows, (int)dim, (int)head_dim, (long)x_stride,
                           (long)dy_stride, (long)dx_stride, eps);
    } else
    hipLaunchKernelGGL(rmsnorm_rope_bwd_kernel, dim3((unsigned)((rows + ROWS_PER_WG - 1) / ROWS_PER_WG)), dim3(ROW_THREADS), 0,
                       (hipStream_t)stream, (const u16*)x, (const u16*)dy, (const u16*)weight, cos_tab, sin_tab, (u16*)dx, dw_acc,
                       (int)rows, (int)dim, (int)head_dim, (long)x_stride, (long)dy_stride, (long)dx_stride, eps);
    GF_CHECK_LAUNCH("gf_rmsnorm_rope_bwd");
    return GF_OK;
}

extern "C" GF_API int gf_colsum(const void* a, int64_t lda, const void* b, int64_t ldb, const void* gate, void* out, int64_t ldo,
                                float* acc, int64_t rows, int64_t cols, void* stream) {
    GF_CHECK_ARG(a && (acc || out), "gf_colsum: null pointer");
    GF_CHECK_ARG((out == nullptr) == (gate == nullptr), "gf_colsum: out and gate go together");
    GF_CHECK_ARG(rows >= 0 && cols > 0 && cols % 8 == 0 && lda % 8 == 0 && (!b || ldb % 8 == 0) && (!out || ldo % 8 == 0),
                 "gf_colsum: cols and leading dimensions must be multiples of 8");
    GF_CHECK_ARG(gf_aligned16(a) && (!b || gf_aligned16(b)) && (!out || gf_aligned16(out)) && (!gate || gf_aligned16(gate)),
                 "gf_colsum: 16-byte alignment required");
    if (rows == 0) return GF_OK;
    const unsigned gx = (unsigned)((cols / 8 + 255) / 256), gy = (unsigned)((rows + 63) / 64);
    hipLaunchKernelGGL(colsum_kernel, dim3(gx, gy), dim3(256), 0, (hipStream_t)stream, (const u16*)a, (const u16*)b,
                       (const u16*)gate, (u16*)out, acc, (int)rows, (int)cols, (long)lda, (long)ldb, (long)ldo);
    GF_CHECK_LAUNCH("gf_colsum");
    return GF_OK;
}

extern "C" GF_API int gf_act_bwd(const void* u, const void* df, void* du, int64_t n, int kind, void* stream) {
    GF_CHECK_ARG(u && df && du, "gf_act_bwd: null pointer");
    GF_CHECK_ARG(n >= 0 && n % 8 == 0 && (kind == 0 || kind == 1), "gf_act_bwd: n must be a multiple of 8, kind 0 (gelu-tanh) or 1 (silu)");
    GF_CHECK_ARG(gf_aligned16(u) && gf_aligned16(df) && gf_aligned16(du), "gf_act_bwd: 16-byte alignment required");
    if (n == 0) return GF_OK;
    hipLaunchKernelGGL(act_bwd_kernel, dim3((unsigned)((n / 8 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const u16*)u,
                       (const u16*)df, (u16*)du, (long)n, kind);
    GF_CHECK_LAUNCH("gf_act_bwd");
    return GF_OK;
}

extern "C" GF_API int gf_mse_loss(const void* pred, const void* target, void* dpred, float* loss, int64_t n, float weight,
                                  void* stream) {
    GF_CHECK_ARG(pred && target && loss, "gf_mse_loss: null pointer");
    GF_CHECK_ARG(n > 0, "gf_mse_loss: empty input");
    hipError_t e = hipMemsetAsync(loss, 0, sizeof(float), (hipStream_t)stream);
    if (e != hipSuccess) {
        gf_set_error("gf_mse_loss: hipMemsetAsync failed: %s", hipGetErrorString(e));
        return GF_ERR_LAUNCH;
    }
    const unsigned blocks = (unsigned)min((long)((n + 255) / 256), 1024L);
    hipLaunchKernelGGL(mse_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const u16*)pred, (const u16*)target,
                       (u16*)dpred, loss, (long)n, weight);
    GF_CHECK_LAUNCH("gf_mse_loss");
    return GF_OK;
}

extern "C" GF_API int gf_adamw_step(void* param, const void* grad, float* exp_avg, float* exp_avg_sq, int64_t n, float lr,
                                    float beta1, float beta2, float eps, float weight_decay, int64_t step, float grad_scale,
                                    void* stream) {
    GF_CHECK_ARG(param && grad && exp_avg && exp_avg_sq, "gf_adamw_step: null pointer");
    GF_CHECK_ARG(n >= 0 && step >= 1, "gf_adamw_step: bad n/step");
    if (n == 0) return GF_OK;
    const float bc1 = 1.0f - powf(beta1, (float)step), bc2s = sqrtf(1.0f - powf(beta2, (float)step));
    hipLaunchKernelGGL(adamw_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (u16*)param,
                       (const u16*)grad, exp_avg, exp_avg_sq, (long)n, lr, beta1, beta2, eps, weight_decay, bc1, bc2s, grad_scale);
    GF_CHECK_LAUNCH("gf_adamw_step");
    return GF_OK;
}

extern "C" GF_API int gf_f32_to_bf16(const float* src, void* dst, int64_t n, void* stream) {
    GF_CHECK_ARG(src && dst && n >= 0, "gf_f32_to_bf16: bad arguments");
    if (n == 0) return GF_OK;
    hipLaunchKernelGGL(f32_to_bf16_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, src, (u16*)dst,
                       (long)n);
    GF_CHECK_LAUNCH("gf_f32_to_bf16");
    return GF_OK;
}

extern "C" GF_API int gf_sumsq(const void* x, int64_t n, float* acc, void* stream) {
    GF_CHECK_ARG(x && acc && n >= 0, "gf_sumsq: bad arguments");
    if (n == 0) return GF_OK;
    const unsigned blocks = (unsigned)min((long)((n + 255) / 256), 2048L);
    hipLaunchKernelGGL(sumsq_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const u16*)x, acc, (long)n);
    GF_CHECK_LAUNCH("gf_sumsq");
    return GF_OK;
}
